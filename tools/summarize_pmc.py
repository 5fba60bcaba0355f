"""Per-kernel means of a rocprofv3 --pmc run:   python tools/summarize_pmc.py <output dir> [kernel name substrings ...]
For every kernel whose name contains one of the substrings (all kernels without any): launches, mean duration, and per
counter the mean over the launches that did work (>= 5 us; all of them if none did), plus the SQ ratios the profiles'
READMEs quote (share of wavefront cycles spent waiting, share of SIMD-busy cycles with the matrix pipe busy)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root, wanted = sys.argv[1], sys.argv[2:]
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("no counter_collection.csv under", root)
        return 1
    acc = defaultdict(lambda: defaultdict(list))
    for fn in files:
        with open(fn) as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"].split("(")[0]
                if wanted and not any(w in name for w in wanted):
                    continue
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
                acc[name][r["Counter_Name"]].append((float(r["Counter_Value"]), dur))
    for name in sorted(acc):
        cs = acc[name]
        means = {}
        any_vals = next(iter(cs.values()))
        work_any = [v for v in any_vals if v[1] >= 5.0] or any_vals
        print("%s: %d launches, %d working, mean %.1f us" % (name, len(any_vals), len(work_any),
                                                              sum(v[1] for v in work_any) / len(work_any)))
        for cname in sorted(cs):
            vals = cs[cname]
            work = [v for v in vals if v[1] >= 5.0] or vals
            means[cname] = sum(v[0] for v in work) / len(work)
            print("    %-28s %16.0f" % (cname, means[cname]))
        if means.get("SQ_WAVE_CYCLES"):
            wc = means["SQ_WAVE_CYCLES"]
            for k in ("SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if k in means:
                    print("    %-28s %15.1f %% of the wavefront cycles" % (k + " share", 100.0 * means[k] / wc))
        if means.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in means:
            print("    %-28s %15.1f %% of the SQ-busy cycles" % ("matrix pipe busy", 100.0 * means["SQ_VALU_MFMA_BUSY_CYCLES"] / means["SQ_BUSY_CYCLES"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
