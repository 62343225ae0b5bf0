"""The C-ABI library loads without a GPU and exports every function include/nkml_hip.h declares."""
import ctypes
import os
import re

from tests.helpers import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "nkml_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nkml_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    from nomalise_kmers_multi_large_amd import binding
    assert declared_functions() == sorted(binding.EXPORTS)


def test_library_exports_every_declared_symbol():
    from nomalise_kmers_multi_large_amd import binding
    lib = binding.load_library()
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert lib.nkml_abi_version() == 5
    assert lib.nkml_strerror(0) == b"ok" and lib.nkml_strerror(-4) == b"sequence byte outside ACGTN"


def test_host_side_entry_points_without_gpu():
    from nomalise_kmers_multi_large_amd import binding
    lib = binding.load_library()
    # keep/skip test is pure host code: float32, strict '<'
    assert lib.nkml_keep_single(0, 0, ctypes.c_float(0.9)) == 1
    assert lib.nkml_keep_single(9, 10, ctypes.c_float(0.9)) == 0
    assert lib.nkml_keep_pair(8, 10, 0, 0, ctypes.c_float(0.9)) == 1
    assert lib.nkml_keep_pair(8, 10, 10, 10, ctypes.c_float(0.9)) == 0
    # argument checks come before any device work
    h = ctypes.c_void_p()
    bad = binding.Config(4, 100, 0, 0, 0, 1024, 1 << 20, 0, 0)
    assert lib.nkml_create(ctypes.byref(bad), ctypes.byref(h)) == binding.NKML_E_ARG
    bad = binding.Config(15, 1, 0, 0, 0, 1024, 1 << 20, 0, 0)
    assert lib.nkml_create(ctypes.byref(bad), ctypes.byref(h)) == binding.NKML_E_ARG


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        return
    import pytest
    import nomalise_kmers_multi_large_amd as nk
    with pytest.raises(nk.NkmlError) as e:
        nk.KmerNormaliser(15, 100)
    assert e.value.code == -6
