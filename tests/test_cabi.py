"""The C-ABI library loads and exports every symbol include/travgpu.h declares (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

from tests.conftest import ROOT


@pytest.fixture(scope="module")
def capi():
    from traversability_estimation_amd import build, capi
    build.build_lib()
    return capi


def test_header_symbols_are_exported(capi):
    hdr = open(os.path.join(ROOT, "include", "travgpu.h")).read()
    declared = set(re.findall(r"\b(te_[a-z_]+)\s*\(", hdr))
    declared -= {"te_ctx"}
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    L = capi.load()
    for s in declared:
        assert hasattr(L, s), s


def test_params_default_and_validation(capi):
    p = capi.default_params()
    assert p.size == C.sizeof(capi.TeParams) and p.normals_radius == 0.05 and p.step_ncrit == 4
    L = capi.load()
    assert L.te_params_validate(C.byref(p)) == 0
    for field, bad, msg in [("slope_critical", 2.0, "Critical slope must be in the interval [0, PI/2]"),
                            ("slope_critical", -0.1, "Critical slope"),
                            ("step_critical", -1.0, "Critical step height"),
                            ("step_radius1", -1.0, "'first_window_radius'"),
                            ("step_radius2", -1.0, "'second_window_radius'"),
                            ("step_ncrit", 0, "'critical_cell_number'"),
                            ("rough_critical", -1.0, "Critical roughness"),
                            ("rough_radius", -1.0, "Roughness estimation radius"),
                            ("normals_axis", 3, "normal_vector_positive_axis")]:
        q = capi.default_params(**{field: bad})
        assert L.te_params_validate(C.byref(q)) == capi.TE_ERR_BAD_PARAM
        assert msg in L.te_last_error().decode()
    q = capi.default_params()
    q.size = 8
    assert L.te_params_validate(C.byref(q)) == capi.TE_ERR_INVALID_ARG


def test_params_blob_roundtrip(capi):
    p = capi.default_params(normals_radius=0.123, step_ncrit=7, w_scale=0.25)
    q = capi.params_from_bytes(capi.params_to_bytes(p))
    assert bytes(q) == bytes(p) and q.step_ncrit == 7


def test_no_cpu_fallback_without_device(capi):
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.TeError) as e:
        capi.Context(0)
    assert e.value.code == capi.TE_ERR_NO_DEVICE


def test_null_arguments_are_rejected_without_a_device(capi):
    """Entry points validate their pointers before they touch the device (status code + te_last_error)."""
    L = capi.load()
    one = (C.c_int * 2)(0, 1)
    xy = (C.c_double * 2)(0.0, 0.0)
    assert L.te_check_footprint_paths(None, 0, 1, one, xy, None, None, None) == capi.TE_ERR_INVALID_ARG
    assert b"te_check_footprint_paths" in L.te_last_error()
    assert L.te_run_chain(None, 0) == capi.TE_ERR_INVALID_ARG
    assert L.te_sync(None) == capi.TE_ERR_INVALID_ARG


def test_header_is_plain_c(tmp_path):
    """include/travgpu.h is the drop-in boundary: it has to compile as C99 and as C++11 on its own."""
    import shutil
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "travgpu.h"\nint main(void) { te_params p; return te_params_default(&p); }\n')
    inc = os.path.join(ROOT, "include")
    for cc, std, extra in (("gcc", "-std=c99", []), ("g++", "-std=c++11", ["-x", "c++"])):
        if shutil.which(cc) is None:
            pytest.skip(f"{cc} not installed")
        subprocess.check_call([cc, std, "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + inc, "-fsyntax-only"] + extra + [str(src)])


def test_shard_range_matches_the_python_split(capi):
    """te_shard_range (what a C++ host uses) and dist.shard_range (what bench.py uses) cut the batch the same way."""
    from traversability_estimation_amd import dist
    for n in (0, 1, 5, 8, 512, 513):
        for world in (1, 2, 3, 8):
            for k in range(world):
                first, count = capi.shard_range(n, world, k)
                assert (first, first + count) == dist.shard_range(n, k, world)
    import ctypes as C
    import pytest
    with pytest.raises(capi.TeError):
        capi.shard_range(8, 0, 0)
    with pytest.raises(capi.TeError):
        capi.shard_range(8, 2, 2)
    L = capi.load()
    assert L.te_bcast_params(None, 2, 0) != 0 and L.te_run_chain_multi(None, 1, 0) != 0 and L.te_sync_multi(None, 1) != 0
    assert L.te_shard_range(4, 2, 0, None, None) != 0
