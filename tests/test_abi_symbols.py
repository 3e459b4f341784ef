"""The C-ABI library loads on a machine without a GPU and exports exactly what include/spock_amd.h declares
(no compute calls here).  Also: the product fails loudly without a GPU -- there is no CPU fallback."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    return True


def _declared():
    txt = open(os.path.join(ROOT, "include", "spock_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(spock_[a-z_A-Z0-9]+)\s*\(", txt)))


def test_header_and_binding_agree(S, built):
    declared = _declared()
    assert "spock_create" in declared and "spock_solve" in declared and len(declared) >= 18
    from spock_jl_amd import _lib
    assert sorted(_lib.SYMBOLS) == declared


def test_library_exports_every_symbol(S, built):
    lib = S.load_library()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.spock_version() == 300 and lib.spock_real_size() == 8
    # the fp32 variant: same sources, same C ABI, real = float
    from spock_jl_amd import _lib
    lib32 = S.load_library(_lib.F32_LIB)
    for name in _declared():
        assert hasattr(lib32, name), name
    assert lib32.spock_real_size() == 4


def test_no_cpu_fallback(S, built):
    """Without a GPU, creating a model raises (SPOCK_ERR_HIP); with one, this test is skipped."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(S.SpockError) as e:
        S.build_model(*S.get_car_fixture())
    assert e.value.code == 3


def test_bad_arguments_are_reported(S, built):
    import ctypes as C
    from spock_jl_amd import _lib
    lib = S.load_library()
    h = C.c_void_p()
    assert lib.spock_create(None, None, C.byref(h)) == _lib.ERR_BAD_ARG
    assert b"NULL" in lib.spock_last_error()
    assert lib.spock_solve(None, None, 1e-3, 1, None, None, 0.1, 0.1, None) == _lib.ERR_BAD_ARG
    assert lib.spock_destroy(None) == 0
