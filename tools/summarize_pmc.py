"""Condenses the rocprofv3 outputs of tools/profile_round.sh into the small files committed under
profiles/<tag>/: kernel_stats_<WL>.csv (copied), pmc_summary_<WL>.json (per kernel and counter:
mean over the launches that did work) and traffic_<WL>.json (HBM bytes per likelihood-kernel
launch; FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md prescribes)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def find(root, suffix):
    hits = glob.glob(os.path.join(root, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def main():
    out, wl = sys.argv[1], sys.argv[2]
    ks = find(os.path.join(out, "kt"), "kernel_stats.csv")
    if ks:
        shutil.copy(ks, os.path.join(out, "kernel_stats_%s.csv" % wl))
    summary = {}
    for d in sorted(glob.glob(os.path.join(out, "pmc*_*"))):
        if not os.path.isdir(d):
            continue
        full = os.path.basename(d).startswith("pmcfull_")
        pruneonly = os.path.basename(d).startswith("pmcprune_")
        cc = find(d, "counter_collection.csv")
        if not cc:
            continue
        acc = defaultdict(lambda: defaultdict(list))
        with open(cc) as f:
            for r in csv.DictReader(f):
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3      # us
                acc[r["Kernel_Name"]][r["Counter_Name"]].append((float(r["Counter_Value"]), dur))
        for k, cs in acc.items():
            for cname, vals in cs.items():
                work = [v for v in vals if v[1] >= 20.0] or vals
                summary.setdefault(k.split("(")[0] + (" [pruning off]" if full else " [no certified stays]" if pruneonly else ""), {})[cname] = {
                    "launches": len(vals), "working_launches": len(work),
                    "mean_working": sum(v[0] for v in work) / len(work),
                    "mean_working_us": sum(v[1] for v in work) / len(work)}
    json.dump(summary, open(os.path.join(out, "pmc_summary_%s.json" % wl), "w"), indent=1, sort_keys=True)
    traffic = {"note": "separate --pmc passes (FETCH_SIZE; WRITE_SIZE), KB units; FETCH_SIZE doubled for "
                       "gfx950 (MI355X_MICROARCH.md); mean over launches >= 20 us"}
    for k, cs in summary.items():
        if "score_mfma" in k and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            if "prune_kernel" in k and "[no certified stays]" in k:
                key = "hbm_bytes_per_launch_pruned"
            elif "prune_kernel" not in k and "[pruning off]" in k:
                key = "hbm_bytes_per_launch"
            else:
                continue
            traffic[key] = int(1024 * (2 * cs["FETCH_SIZE"]["mean_working"] + cs["WRITE_SIZE"]["mean_working"]))
            traffic[key + "_kernel"] = k
            traffic[key + "_fetch_kb_raw"] = round(cs["FETCH_SIZE"]["mean_working"], 1)
            traffic[key + "_write_kb"] = round(cs["WRITE_SIZE"]["mean_working"], 1)
    for k, cs in summary.items():
        if k.startswith("certify_kernel") and "[" not in k and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            traffic["hbm_bytes_per_launch_certified"] = int(1024 * (2 * cs["FETCH_SIZE"]["mean_working"]
                                                                    + cs["WRITE_SIZE"]["mean_working"]))
            traffic["hbm_bytes_per_launch_certified_fetch_kb_raw"] = round(cs["FETCH_SIZE"]["mean_working"], 1)
            traffic["hbm_bytes_per_launch_certified_write_kb"] = round(cs["WRITE_SIZE"]["mean_working"], 1)
    json.dump(traffic, open(os.path.join(out, "traffic_%s.json" % wl), "w"), indent=1, sort_keys=True)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
