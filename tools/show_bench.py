import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d["roofline"]; print(f, d["value"], d["ms_per_step"], "live", r["avg_launch_ms"], r["frac"], "alone", r["alone_on_the_gpu"]["avg_launch_ms"], r["alone_on_the_gpu"]["frac"], "resident", d["extra"].get("resident_inputs_sweeps_per_s"))
    except Exception as e: print(f, "ERR", e)
