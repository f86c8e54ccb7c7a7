"""The microbenchmarks under tools/ubench are cited as evidence in DESIGN.md (issue model, residency, range check): keep them compiling."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("name", ["mfma_issue", "residency", "soffset_probe"])
def test_microbenchmark_cross_compiles_for_gfx950(name, tmp_path):
    src = os.path.join(ROOT, "tools", "ubench", name + ".hip")
    out = tmp_path / (name + ".o")
    r = subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", "-Wno-unused-value", "-c", src, "-o", str(out)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.stat().st_size > 0
