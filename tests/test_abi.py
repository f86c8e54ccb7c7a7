"""CPU-side checks of the boundary: the C-ABI library loads without a GPU, exports every function include/dd_hotpath.h
declares, the ctypes table mirrors the header one to one, and the host-side validation refuses bad arguments before any
launch (no compute calls here)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dd_hotpath.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from driving_dirty_amd import _lib
    from driving_dirty_amd.build import LIB
    assert os.path.exists(LIB), "build the HIP library first (python __graft_entry__.py)"
    handle = ctypes.CDLL(LIB)
    names = declared_functions()
    assert len(names) >= 40
    for n in names:
        assert hasattr(handle, n), f"{n} is declared in include/dd_hotpath.h but not exported by the library"
    # the ctypes signature table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == names
    assert _lib.lib().dd_abi_version() == _lib.ABI_VERSION


def test_descriptor_structs_match_the_header():
    from driving_dirty_amd import _lib
    text = open(HEADER).read()
    for struct, cls in (("dd_conv_desc", _lib.ConvDesc), ("dd_gconv_desc", _lib.GConvDesc)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), text, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = [f.strip() for decl in re.findall(r"int32_t([^;]*);", body) for f in decl.split(",")]
        assert fields == [n for n, _ in cls._fields_], struct
        assert ctypes.sizeof(cls) == 4 * len(fields)


def test_host_side_validation_without_a_gpu():
    from driving_dirty_amd import _lib, ops
    lib = _lib.lib()
    d = _lib.ConvDesc(2, 16, 22, 32, 32, 32, 3, 1, 1, 0)
    assert lib.dd_conv_packed_floats(ctypes.byref(d), 0) == 36 * 64 * 4
    assert lib.dd_conv_wgrad_workspace_bytes(ctypes.byref(d)) > 0
    bad = _lib.ConvDesc(2, 16, 22, 32, 32, 32, 5, 1, 2, 0)
    assert lib.dd_conv_packed_floats(ctypes.byref(bad), 0) == -1
    assert b"k3 p1" in lib.dd_last_error()
    assert lib.dd_set_cu_budget(0) != 0 and lib.dd_set_cu_budget(256) == 0
    # NULL pointers / bad sizes are refused by the entry points themselves
    assert lib.dd_stitch6(None, None, None, None, 1, 4, 4, -1, None) == 2
    assert lib.dd_linear_fwd(None, None, None, None, 4, 8, 6, None, 0, None) != 0          # K % 4 != 0 -> unsupported first
    assert lib.dd_adam_step(None, None, None, None, 16, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0, None) == 2
    assert lib.dd_adam_step_multi(None, 0, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0, None) == 2
    # the Python shims refuse CPU tensors loudly: there is no CPU fallback
    with pytest.raises(_lib.HotpathError):
        ops.pool4_fwd(torch.zeros(1, 4, 4, 32))
    from driving_dirty_amd.components import Encoder
    with pytest.raises(RuntimeError):
        Encoder(16, 8, 3, 16, 22)(torch.zeros(2, 3, 16, 22))


def test_missing_library_fails_loudly(monkeypatch):
    from driving_dirty_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB", "/nonexistent/libdd_hotpath.so")
    with pytest.raises(_lib.HotpathError):
        _lib.lib()
