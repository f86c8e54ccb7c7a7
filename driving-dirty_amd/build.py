"""Build the C-ABI shared library (``csrc/libdd_hotpath.so``) with hipcc for gfx950.

    python -m driving_dirty_amd.build          # or __graft_entry__.build()

hipcc cross-compiles without a GPU.  The library is built IN-TREE so that it travels with the
repository snapshot to the GPU box (a JIT cache under ~/.cache would not).  Each source is compiled
to its own object (in parallel, re-done only when the source or a header is newer) and the objects
are linked into the one library.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libdd_hotpath.so")
OBJ = os.path.join(CSRC, "build")
SOURCES = ["runtime.hip", "conv3x3.hip", "layout_pool.hip", "dense.hip", "linear.hip", "gconv.hip", "dconv.hip", "dconv_t.hip", "dconv_m.hip", "dconv_split.hip", "conv1ch.hip", "ssconv.hip", "bn2d.hip", "raster.hip",
           "conv3x3_bf16.hip", "mlp_tail.hip", "adam_rankb.hip", "strip6.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# per-file additions.  adam_rankb.hip: MFMA accumulators in ordinary vector registers (the kernel's whole budget is 72 registers, and the
# default form keeps a second copy of the accumulators in the accumulation registers)
EXTRA_FLAGS = {"adam_rankb.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
HEADERS = [os.path.join(CSRC, "dd_common.h"), os.path.join(CSRC, "dd_adam.h"), os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "dd_hotpath.h")]


def _obj(src):
    return os.path.join(OBJ, os.path.splitext(src)[0] + ".o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build():
    return _stale(LIB, [os.path.join(CSRC, s) for s in SOURCES] + HEADERS)


def build_diag(verbose=True):
    """``csrc/libdd_hotpath_diag.so``: the same library compiled with -DDD_TIMING_DIAG (timing ablations of single kernels whose
    results are then WRONG: DD_DCONV_REPEAT, DD_SPLIT_ABL).  Never loaded unless DD_HOTPATH_LIB points at it; not built by build()."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = os.path.join(CSRC, "build_diag")
    os.makedirs(out, exist_ok=True)

    def one(src):
        obj = os.path.join(out, os.path.splitext(src)[0] + ".o")
        if _stale(obj, [os.path.join(CSRC, src)] + HEADERS):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-DDD_TIMING_DIAG", "-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return obj
    with ThreadPoolExecutor(max_workers=8) as pool:
        objs = list(pool.map(one, SOURCES))
    lib = os.path.join(CSRC, "libdd_hotpath_diag.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    todo = [s for s in SOURCES if force or _stale(_obj(s), [os.path.join(CSRC, s)] + HEADERS)]

    def compile_one(src):
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(todo)))) as pool:
        list(pool.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [_obj(s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    if "--diag" in sys.argv:
        print(build_diag())
    else:
        build(force="--force" in sys.argv)
