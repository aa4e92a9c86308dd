"""Build libmaai_hip.so (gfx950) in-tree with hipcc.  No torch, no cmake.

    python multimodal-active-ai_amd/build.py [--force]

Objects go to build/ (git-ignored), the library to lib/libmaai_hip.so (git-ignored,
but it travels to the GPU box with the gpurun snapshot).
"""
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "lib", "libmaai_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X library cannot be built")
    return exe


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def newest_dep():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "maai_hip.h")]
    return max(os.path.getmtime(p) for p in deps)


def compile_one(src):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    live = set(os.path.basename(p)[:-4] + ".o" for p in sources())
    for f in os.listdir(OBJ):   # objects of sources that no longer exist
        if f.endswith(".o") and f not in live:
            try:
                os.remove(os.path.join(OBJ, f))
            except OSError:
                pass
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h"))
    hdr_time = max(hdr_time, os.path.getmtime(os.path.join(HERE, "..", "include", "maai_hip.h")))
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_time):
        return obj, ""
    r = subprocess.run([hipcc(), *FLAGS, "-c", src, "-o", obj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr))
    return obj, r.stderr


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) > newest_dep():
        return LIB
    with cf.ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        res = list(ex.map(compile_one, sources()))
    objs = [o for o, _ in res]
    if verbose:
        for _, err in res:
            if err.strip():
                sys.stderr.write(err)
    tmp = "%s.%d.tmp" % (LIB, os.getpid())   # never expose a half-written library to a concurrent loader
    r = subprocess.run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("link failed:\n" + r.stderr)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
