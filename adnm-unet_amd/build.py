"""Builds libadnm_hip.so (gfx950) in-tree from csrc/*.hip with hipcc.  No cmake, no JIT cache:
the .so lands next to adnm_hip/lib.py so it travels with the source tree."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "adnm_hip", "libadnm_hip.so")
FLAGS = ["--offload-arch=gfx950", os.environ.get("ADNM_HIPCC_OPT", "-O3"), "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(src, dst, deps=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in (src,) + tuple(deps))


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    deps = tuple(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))) + (
        os.path.join(os.path.dirname(HERE), "include", "adnm_hip.h"),)
    jobs = []
    for f in srcs:
        src, obj = os.path.join(CSRC, f), os.path.join(OBJ, f[:-4] + ".o")
        if force or _newer(src, obj, deps):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[adnm build] compiled {os.path.basename(src)}", flush=True)
        return obj

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, f[:-4] + ".o") for f in srcs]
    if force or jobs or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[adnm build] linked {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
