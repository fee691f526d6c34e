"""Periodic tree-only path (PERIODIC without PMGRID): Ewald / lattice-sum correction tables, the correction walk
(forcetree.c:2077-2455) and the PERIODIC direct sum with lattice_corr (forcetree.c:3515-3529)."""
import numpy as np
import pytest

from conftest import rel_err
from ewald import ewald_direct


def _case(pkg, wiring, ng, n=2500, L=100.0, seed=3, **kw):
    pos, mass, typ = pkg.ic.uniform_box(n, box=L, n_gravs=ng, seed=seed)
    eps = L / (40 * n ** (1 / 3))
    cfg = pkg.make_config(n_gravs=ng, periodic=1, pmgrid=0, box_size=L, G=1.0, theta=0.5, softening=[eps] * 6,
                          type_to_grav=pkg.ic.default_type_to_grav(ng), wiring=wiring, **kw)
    return cfg, pos, mass, typ, eps


def _truth(pkg, cfg, pos, mass, typ, idx, eps):
    ng = cfg.n_gravs
    species = np.array(pkg.ic.default_type_to_grav(ng))[typ]
    law = [[cfg.law_accel[i][j] for j in range(ng)] for i in range(ng)]
    return ewald_direct(pos, mass, species, idx, cfg.box_size, cfg.G, law, cfg.yukawa_imass / cfg.box_size, 2.8 * eps)


@pytest.mark.parametrize("wiring,ng", [("newton", 1), ("c4", 2)])
def test_oracle_lattice_tables_against_independent_ewald(pkg, O, wiring, ng):
    """the restated ewald_force / yukawa_lattice_force tables + trilinear lattice_corr reproduce a textbook Ewald sum
    to the interpolation error of the 65^3 table"""
    cfg, pos, mass, typ, eps = _case(pkg, wiring, ng)
    lat = O.lattice_tables(cfg)
    assert np.all(lat[:, :, :, 0, 0, 0] == 0)                       # no self-force from the own images
    idx = np.arange(0, len(pos), 25, dtype=np.int32)
    d = O.direct_lattice(cfg, pos, mass, typ, idx, lat)
    e = rel_err(d, _truth(pkg, cfg, pos, mass, typ, idx, eps))
    assert e.max() < 1e-4 and np.median(e) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("wiring,ng", [("newton", 1), ("c4", 2)])
def test_gpu_periodic_tree_only(pkg, O, wiring, ng):
    cfg, pos, mass, typ, eps = _case(pkg, wiring, ng, n=6000, walk_mode=pkg.WALK_STRICT)
    lat = O.lattice_tables(cfg)
    idx = np.arange(0, len(pos), 30, dtype=np.int32)
    truth = _truth(pkg, cfg, pos, mass, typ, idx, eps)
    eng = pkg.Engine(cfg)
    eng.set_particles(pos, mass, typ)
    eng.compute_accelerations(pm_step=False)
    acc, old, cost = eng.get_accel()
    # GPU lattice tables + lattice_corr == oracle's
    d_gpu = eng.direct_sum(idx)
    assert rel_err(d_gpu, O.direct_lattice(cfg, pos, mass, typ, idx, lat)).max() < 1e-9
    # strict walk + lattice-correction walk == oracle (force_treeevaluate + force_treeevaluate_lattice_correction)
    T = O.Tree(cfg, pos, mass, typ)
    a_o, n_o = T.walk()
    a_o, n_o = O.lattice_walk(T, a_o, n_o, lat)
    a_o, old_o = O.finish(cfg, a_o)
    assert np.abs(acc - a_o).max() / np.abs(a_o).max() < 1e-9
    assert np.array_equal(cost.astype(np.int64), n_o)
    # relative-criterion pass, both walks, against the independent Ewald truth
    eng.set_opening(0.0, 0.005)
    eng.set_old_acc(old)
    eng.gravity_tree()
    acc2, _, _ = eng.get_accel()
    cfg.err_tol_theta = 0.0
    a2, n2 = T.walk(old_acc=old_o)
    a2, n2 = O.lattice_walk(T, a2, n2, lat, old_acc=old_o)
    a2, _ = O.finish(cfg, a2)
    assert np.abs(acc2 - a2).max() / np.abs(a2).max() < 1e-8
    eng.set_walk_mode(pkg.WALK_GROUP)
    eng.gravity_tree()
    acc_g, _, cost_g = eng.get_accel()
    rms = lambda e: float(np.sqrt(np.mean(e ** 2)))
    e_ref, e_grp = rel_err(a2[idx], truth), rel_err(acc_g[idx], truth)
    print("periodic tree-only vs Ewald [%s]: reference walk rms %.2e, group walk rms %.2e" % (wiring, rms(e_ref), rms(e_grp)))
    assert rms(e_grp) <= rms(e_ref) * 1.05
    eng.close()


@pytest.mark.gpu
def test_gpu_periodic_direct_sum_is_the_treepm_truth(pkg):
    """with PERIODIC the GPU direct sum includes lattice_corr, i.e. it is what FORCETEST compares TreePM against
    (gravtree_forcetest.c:297-311): the TreePM total must agree with it like it does with the Ewald golden"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_ewald_golden import N, L, SEED, case_config
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "ewald_truth_c4.npz"))
    pos, mass, typ = pkg.ic.uniform_box(N, box=L, n_gravs=2, seed=SEED)
    cfg, eps = case_config(pkg, "c4", 2, walk_mode=pkg.WALK_GROUP)
    eng = pkg.Engine(cfg)
    eng.set_particles(pos, mass, typ, old_acc=gold["old_acc"])
    eng.set_opening(0.0, 0.005)
    eng.compute_accelerations(pm_step=True)
    acc, _, _, pm = eng.get_accel(want_pm=True)
    idx = gold["idx"].astype(np.int32)
    d = eng.direct_sum(idx)
    assert rel_err(d, gold["truth"]).max() < 2e-4                    # GPU Ewald-corrected direct sum == independent Ewald
    e = rel_err((acc + pm)[idx], d)
    assert float(np.sqrt(np.mean(e ** 2))) < 1e-2
    eng.close()
