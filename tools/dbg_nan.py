import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import importlib
pkg = importlib.import_module("gadget-2.0.7-ngravs_amd")
from make_ewald_golden import N, L, SEED, case_config
gold = np.load("tests/golden/ewald_truth_c4.npz")
pos, mass, typ = pkg.ic.uniform_box(N, box=L, n_gravs=2, seed=SEED)
for tune in ({}, {"walk_ring": 0}, {"walk_ring_k": 4}):
    cfg, eps = case_config(pkg, "c4", 2, walk_mode=pkg.WALK_GROUP, group_reach=0.0)
    eng = pkg.Engine(cfg)
    for k, v in tune.items():
        eng.set_tuning(**{k: v})
    eng.set_particles(pos, mass, typ, old_acc=gold["old_acc"])
    eng.set_opening(0.0, 0.005)
    eng.compute_accelerations(pm_step=True)
    acc, _, cost, gpm = eng.get_accel(want_pm=True)
    bad = ~np.isfinite(acc).all(axis=1)
    print(tune, "N", N, "nan rows", bad.sum(), "first", np.nonzero(bad)[0][:10], "cost mean", cost.mean(), "type of bad", np.bincount(typ[bad]) if bad.any() else None)
    if bad.any():
        i = np.nonzero(bad)[0]
        print(" groups of bad rows (peano order unknown) count per 64-block of index:", len(np.unique(i // 64)))
    eng.close()
