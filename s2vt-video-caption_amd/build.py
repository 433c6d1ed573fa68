"""Build libs2vt_hip.so (gfx950) in-tree with hipcc.  No torch involvement: the library is a plain
C-ABI shared object (include/s2vt_hip.h) that only links the HIP runtime."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libs2vt_hip.so")
SOURCES = ["api_runtime.hip", "api_shared.hip", "api_train.hip", "api_decode.hip", "api_beam.hip", "api_ops.hip", "options.hip", "gemm.hip", "gemm_bf16.hip", "gemm_x3.hip", "gemm_b1.hip", "split.hip", "lstm.hip", "lstm_gemv.hip", "lstm_bf16.hip", "lstm_persist.hip", "lstm_persist_x3.hip", "argmax_x3.hip", "ce.hip", "misc.hip", "beam_queue.hip"]
HEADERS = ["common.h", "kernels.h", "experiment.h", "api_internal.h", os.path.join("..", "..", "include", "s2vt_hip.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, defines=(), out_path=None):
    """Compile every HIP source for gfx950 and link the shared library. Returns its path.
    `defines`/`out` build an experimental variant (the in-kernel stamp tools, tools/bench_*_stamps.py) next to the product library."""
    if out_path is None and not force and not needs_build():
        return LIB
    objdir = os.path.join(HERE, "build" if out_path is None else "build_" + os.path.basename(out_path))
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fno-gpu-rdc", "-Wall",
             "-Wno-unused-function", "-Wno-unused-variable"] + ["-D" + d for d in defines]
    procs = []
    for s in SOURCES:
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        cmd = [hipcc] + flags + ["-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    objs = []
    for s, obj, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (s, out.decode(errors="replace")))
        if verbose and out:
            print(out.decode(errors="replace"))
        objs.append(obj)
    target = LIB if out_path is None else out_path
    tmp = target + ".tmp"
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", tmp] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stdout.decode(errors="replace"))
    os.replace(tmp, target)
    return target


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
