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


def test_exports_the_c_host_layer(pkg, have_lib):
    hdr = open(os.path.join(os.path.dirname(pkg.__file__), "..", "include", "ngravs_host.h")).read()
    declared = set(re.findall(r"\b(ngravs_host_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(pkg.HOST_EXPORTS)
    for name in declared:
        assert hasattr(have_lib, name), name


def test_host_split_balances_work_within_the_memory_bound(pkg, have_lib):
    """ngravs_host_split = domain_findSplit (by count, under max_load) + domain_shiftSplit (by work), reference
    domain.c:347-544: contiguous runs of cells, every task gets cells, the work maximum never exceeds the count-only cut's,
    and a clustered work distribution is balanced far better than by count."""
    import numpy as np
    rng = np.random.default_rng(3)
    ncell, ntask = 4096, 8
    count = rng.integers(50, 150, ncell).astype(np.int64)
    work = count.astype(np.float64)
    work[1000:1200] *= 40.0                         # a clump: few particles more, many interactions more
    owner_c = np.zeros(ncell, dtype=np.int32)
    owner_w = np.zeros(ncell, dtype=np.int32)
    maxload = 1.5 * count.sum() / ntask
    assert have_lib.ngravs_host_split(count.ctypes.data, None, ncell, ntask, maxload, owner_c.ctypes.data) == 0
    assert have_lib.ngravs_host_split(count.ctypes.data, work.ctypes.data, ncell, ntask, maxload, owner_w.ctypes.data) == 0
    for owner in (owner_c, owner_w):
        assert np.all(np.diff(owner) >= 0) and owner[0] == 0 and owner[-1] == ntask - 1      # contiguous runs, in task order
        assert len(np.unique(owner)) == ntask
        assert np.bincount(owner, weights=count).max() <= maxload
    wc = np.bincount(owner_c, weights=work)
    ww = np.bincount(owner_w, weights=work)
    cc = np.bincount(owner_c, weights=count)
    assert cc.max() / cc.mean() < 1.02                                                      # count cut: balanced by count
    assert ww.max() <= wc.max() and ww.max() / ww.mean() < 0.75 * wc.max() / wc.mean()       # work cut: better by work
    # an impossible memory bound is reported, not silently violated
    assert have_lib.ngravs_host_split(count.ctypes.data, None, ncell, ntask, 0.5 * count.sum() / ntask, owner_c.ctypes.data) == -1


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
