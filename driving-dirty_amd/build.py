"""Build the C-ABI shared library (``csrc/libdd_hotpath.so``) with hipcc for gfx950.

    python -m driving_dirty_amd.build          # or __graft_entry__.build()

hipcc cross-compiles without a GPU.  The library is built IN-TREE so that it travels with the
repository snapshot to the GPU box (a JIT cache under ~/.cache would not).
"""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libdd_hotpath.so")
SOURCES = ["runtime.hip", "conv3x3.hip", "layout_pool.hip", "dense.hip", "linear.hip", "gconv.hip", "bn2d.hip", "raster.hip", "conv3x3_bf16.hip", "mlp_tail.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "dd_common.h"),
            os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "dd_hotpath.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
