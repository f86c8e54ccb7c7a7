"""No register spills in the shipped kernels: read from the AMDGPU metadata of the gfx950 code objects embedded in the built
library (tools/kernel_resources.py; no GPU, no recompilation).  A kernel that uses the whole register file and then spills
inside its MFMA loop loses more than the occupancy it bought -- and DESIGN.md rejects variants for exactly that, so the shipped
ones are held to it too."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_kernel_spills_registers():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import kernel_resources
    finally:
        sys.path.pop(0)
    ks = kernel_resources.kernels()
    assert len(ks) > 150
    spilled = {k[".name"]: (k.get(".vgpr_spill_count", 0), k.get(".sgpr_spill_count", 0)) for k in ks if k.get(".vgpr_spill_count", 0)}
    assert not spilled, f"VGPR spills: {spilled}"
    # scratch (private segment) at all: only the box rasteriser, whose per-thread span list is a genuinely indexed local array
    scratch = {k[".name"]: k[".private_segment_fixed_size"] for k in ks if k[".private_segment_fixed_size"]}
    assert all("raster_kernel" in n for n in scratch), scratch
    # the kernels VERDICT r02 named
    by = {k[".name"]: k for k in ks}
    for frag in ("conv_wino2_wgradILi4", "dconv_fwd_kernelILi7ELi7ELi2", "dconv_fwd_kernelILi7ELi3ELi2", "dconv_fwd_kernelILi6ELi6ELi2",
                 "dconv_fwd_kernelILi8ELi8ELi2"):
        hit = [k for n, k in by.items() if frag in n]
        assert hit and all(k[".private_segment_fixed_size"] == 0 for k in hit), frag


def test_rankb_optimizer_kernels_fit_beside_the_c2_weight_gradient():
    """dd_adam_step_rankb runs on the optimizer's side stream beside conv_wino2_wgrad<4> (440 registers, 133 KB of LDS per workgroup, one
    wave per SIMD on every CU): what is left of a SIMD's 512 registers is 72, of a CU's 160 KB of LDS 27 KB.  A variant that needs more
    does not become resident until the conv kernel has drained (measured: +1.6 ms on a 7.6 ms step with a 76-register build)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import kernel_resources
    finally:
        sys.path.pop(0)
    # (adam_rankb_wide_kernel is the form for the pass that runs BY ITSELF -- more than one workgroup per CU -- and has no budget to keep)
    ks = [k for k in kernel_resources.kernels() if "adam_rankb" in k[".name"] and "wide" not in k[".name"]]
    assert len(ks) == 3, [k[".name"] for k in ks]
    wgrad = [k for k in kernel_resources.kernels() if "conv_wino2_wgradILi4" in k[".name"]]
    assert wgrad and all(k[".vgpr_count"] <= 440 for k in wgrad)
    for k in ks:
        assert k[".vgpr_count"] + k.get(".agpr_count", 0) <= 72, (k[".name"], k[".vgpr_count"], k.get(".agpr_count"))
        assert k[".group_segment_fixed_size"] <= 24 * 1024, (k[".name"], k[".group_segment_fixed_size"])
