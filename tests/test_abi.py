"""The C-ABI library loads on a CPU-only box and exports every symbol include/ngravs_hip.h declares."""
import ctypes as C
import os
import re


def test_exports_every_declared_symbol(pkg, have_lib):
    hdr = open(os.path.join(os.path.dirname(pkg.__file__), "..", "include", "ngravs_hip.h")).read()
    declared = set(re.findall(r"\b(ngravs_[a-z0-9_]+)\s*\(", hdr)) - {"ngravs_fatal_fn"}
    assert declared, "no declarations found"
    assert declared == set(pkg.EXPORTS)
    for name in declared:
        assert hasattr(have_lib, name), name


def test_struct_sizes_match_header(pkg, have_lib):
    info = have_lib.ngravs_build_info().decode()
    sizes = dict(re.findall(r"sizeof\((\w+)\)=(\d+)", info))
    assert int(sizes["config"]) == C.sizeof(pkg.Config)
    assert int(sizes["particles"]) == C.sizeof(pkg.Particles)
    assert int(sizes["stats"]) == C.sizeof(pkg.Stats)
    assert have_lib.ngravs_abi_version() == pkg.abi.ABI_VERSION


def test_bad_wiring_is_rejected_without_a_gpu(pkg, have_lib):
    cfg = pkg.make_config(n_gravs=2, wiring="newton")
    cfg.law_accel[0][1] = pkg.LAW_YUKAWA          # asymmetric: violates Newton's third law probe
    h = C.c_void_p()
    assert have_lib.ngravs_create(C.byref(cfg), C.byref(h)) == -6
    cfg = pkg.make_config(n_gravs=1, pmgrid=32, periodic=0, box_size=1.0)
    assert have_lib.ngravs_create(C.byref(cfg), C.byref(h)) == -1   # non-periodic PM is disabled by ngravs
    cfg = pkg.make_config(n_gravs=1)
    cfg.n_gravs = 4
    assert have_lib.ngravs_create(C.byref(cfg), C.byref(h)) == -1
