"""In-tree build of librecengine.so (hipcc, gfx950 only).

    python -m paddlerec_amd.build            # incremental
    python -m paddlerec_amd.build --force

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
paddlerec_amd/librecengine.so is git-ignored but travels to the GPU box with the snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "librecengine.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(REPO, "include"),
         "-I" + CSRC, "-Wno-unused-result"]
FLAGS += os.environ.get("REC_HIPCC_DEFINES", "").split()      # measurement builds only (e.g. -DREC_DIN_PHASE_TIMING)


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(REPO, "include", "recengine.h"))
    jobs = []
    for s in sources():
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s[:-4] + ".o")
        if force or _newer(src, obj) or any(_newer(h, obj) for h in headers):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj, "-Rpass-analysis=kernel-resource-usage"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        with open(obj[:-2] + ".resources.txt", "w") as f:       # registers / occupancy / LDS of every kernel of this
            f.write(r.stderr)                                   # object (tests/test_kernel_resources.py reads it)
        return src

    if jobs:
        if verbose:
            print("[build] hipcc %s: %s" % (ARCH, ", ".join(os.path.basename(j[0]) for j in jobs)),
                  flush=True)
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in sources()]
    if jobs or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
        if verbose:
            print("[build] linked", LIB, flush=True)
    try:          # the custom-op shim is an optional integration layer: a box without a host C++ compiler still gets
        build_paddle_ops(force or bool(jobs), verbose)            # librecengine.so (cpp_extension.load raises "not built")
    except (OSError, RuntimeError) as e:
        print("[build] WARNING: custom-op shim not built (%s)" % str(e).splitlines()[0][:200], flush=True)
    return LIB


PADDLE_OPS = os.path.join(HERE, "paddle_ops")
PADDLE_OPS_LIB = os.path.join(PADDLE_OPS, "librec_paddle_ops.so")


def build_paddle_ops(force=False, verbose=True):
    """The Paddle custom-op shim (paddle_ops/rec_paddle_ops.cc): host C++ only, compiled against the executable stand-in
    of paddle/extension.h (paddle_ops/mock) because PaddlePaddle is not installable here, linked to librecengine.so."""
    src = os.path.join(PADDLE_OPS, "rec_paddle_ops.cc")
    deps = [src, os.path.join(PADDLE_OPS, "mock", "paddle", "extension.h"), os.path.join(REPO, "include", "recengine.h")]
    if not (force or any(_newer(d, PADDLE_OPS_LIB) for d in deps)):
        return PADDLE_OPS_LIB
    import shutil
    cxx = os.environ.get("CXX") or ("g++" if shutil.which("g++") else _hipcc())      # hipcc compiles host C++ as well
    cmd = [cxx, "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-Wno-unused-parameter", "-fvisibility=hidden",
           "-I" + os.path.join(PADDLE_OPS, "mock"), "-I" + os.path.join(REPO, "include"), src, "-o", PADDLE_OPS_LIB,
           "-L" + HERE, "-lrecengine", "-Wl,-rpath,$ORIGIN/.."]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("custom-op shim failed to compile:\n" + r.stderr[-6000:])
    if verbose:
        print("[build] linked", PADDLE_OPS_LIB, flush=True)
    return PADDLE_OPS_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
