"""The C-ABI library loads and exports every symbol include/admm_hip.h declares (no compute)."""
import os
import re

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "admm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(admm_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    from admm_amd import build, capi
    build.build_lib()
    L = capi.load()
    declared = _declared()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), name
    assert sorted(capi.SYMBOLS) == declared
    assert L.admm_abi_version() == 1


def test_fails_loudly_without_gpu():
    import ctypes as C
    import torch
    from admm_amd import capi
    if torch.cuda.is_available():
        return
    L = capi.load()
    h = C.c_void_p()
    rc = L.admm_ctx_create(0, None, C.byref(h))
    assert rc != 0 and b"no CPU path" in L.admm_last_error()
    try:
        capi.Context(0)
    except capi.AdmmError:
        pass
    else:
        raise AssertionError("Context() must raise without a GPU")
