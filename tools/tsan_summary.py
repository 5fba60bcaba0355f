"""Sorts a ThreadSanitizer log of tools/tsan_run.py:   python tools/tsan_summary.py gpurun_out/r06/tsan_host.log
A report concerns the library when the ACCESSING frame (the first frame that is not an interceptor of the sanitizer) of
either side lies in libbgmm_hip_tsan.so; reports whose accesses are both inside the HIP / HSA runtimes (not instrumented:
the sanitizer does not see their own synchronisation) are counted per runtime entry and set aside."""
import collections
import re
import sys

text = open(sys.argv[1], errors="replace").read()
reports = [r for r in text.split("==================") if "WARNING: ThreadSanitizer" in r]
ours, theirs = [], collections.Counter()
for r in reports:
    kind = re.search(r"WARNING: ThreadSanitizer: ([^\(]+)", r).group(1).strip()
    sides = re.split(r"\n  (?=(?:Previous |)(?:Atomic |)(?:Write|Read|write|read)|Mutex|Thread T|Location)", r)
    acc = []
    for s in sides:
        if not re.match(r"(?:Previous |)(?:Atomic |)(?:Write|Read|write|read)", s.strip()):
            continue
        frames = re.findall(r"#\d+ (\S+) .*?\(([^+\)]+)\+0x[0-9a-f]+\)", s)
        first = next(((fn, lib) for fn, lib in frames if "tsan" not in lib), ("?", "?"))
        acc.append(first)
    if any("libbgmm_hip" in lib for _, lib in acc):
        ours.append((kind, acc, r))
    else:
        theirs[(kind, tuple(sorted(set(lib for _, lib in acc))))] += 1
print("%d reports; %d with an accessing frame in libbgmm_hip_tsan.so" % (len(reports), len(ours)))
for (kind, libs), n in theirs.most_common():
    print("  %4d  %-28s accesses inside %s" % (n, kind, ", ".join(libs)))
for kind, acc, r in ours:
    print("---- %s: %s" % (kind, acc))
    print("\n".join(r.strip().split("\n")[:40]))
