"""
Build libbgmm_hip.so (hipcc, gfx950 only) in-tree next to its sources.

The shared object is git-ignored but travels to the GPU box with the gpurun
snapshot.  ``build()`` is idempotent: it recompiles only sources newer than
their objects.
"""
import glob
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbgmm_hip.so")
SOURCES = ["api_context.hip", "api_inputs.hip", "api_perm.hip", "api_sweep.hip", "api_group.hip", "api_comm.hip", "kernels_state.hip", "kernels_score.hip", "kernels_prune.hip", "kernels_choice.hip",
           "kernels_resolve.hip", "kernels_rng.hip", "kernels_seq.hip", "kernels_gram.hip", "kernels_home.hip", "kernels_safe.hip",
           "kernels_perm.hip", "kernels_resid.hip"]
# every header next to the sources + the public one: editing any of them rebuilds every object
HEADERS = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(os.path.dirname(HERE), "include", "bgmm.h")]
# -ffp-contract=off: the sufficient-statistics updates must round product and sum
# separately (bit-identical m / S to the reference); hot loops call fma() explicitly.
# -fvisibility=hidden: the shared object exports the BGMM_API entry points of include/bgmm.h and nothing else.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-result"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libbgmm_hip.so cannot be built")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=()):
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src + ".o")
        if force or _stale(o, [s] + HEADERS):
            jobs.append([cc] + FLAGS + list(extra_flags) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stdout))
        return r.stdout

    if jobs:
        with ThreadPoolExecutor(max_workers=4) as ex:
            outs = list(ex.map(run, jobs))
        if verbose:
            for o in outs:
                if o.strip():
                    print(o)
    objs = [os.path.join(objdir, s + ".o") for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
