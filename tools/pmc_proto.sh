export TMPDIR=/tmp
cd /tmp
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU"; do
rm -rf /tmp/pp; rocprofv3 --pmc $C --kernel-trace -d /tmp/pp -o p --output-format csv -- $GRAFT_REPO_ROOT/tools/home_proto > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/pp/**/*counter_collection.csv",recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
for k,v in acc.items():
    print(k)
    for c,vals in sorted(v.items()):
        vals=vals[len(vals)//2:]
        print("   %-28s %16.0f   avg over %d launches, %.1f us" % (c, sum(x[0] for x in vals)/len(vals), len(vals), sum(x[1] for x in vals)/len(vals)*1e-3))
PY
done
