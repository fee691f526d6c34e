#!/usr/bin/env python3
"""GPU: accuracy (vs the Ewald golden) and cost of the group walk as a function of group_reach."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import __graft_entry__ as ge
from make_ewald_golden import N, L, SEED, CASES, case_config
pkg = ge.load_package()
rms = lambda e: float(np.sqrt(np.mean(e ** 2)))
for wiring, ng in CASES.items():
    gold = np.load(os.path.join(ROOT, "tests", "golden", "ewald_truth_%s.npz" % wiring))
    pos, mass, typ = pkg.ic.uniform_box(N, box=L, n_gravs=ng, seed=SEED)
    idx, truth = gold["idx"], gold["truth"]
    e = np.linalg.norm(gold["ref_total"] - truth, axis=1) / np.linalg.norm(truth, axis=1)
    print("%s reference walk: rms %.3e max %.2e ia %.1f" % (wiring, rms(e), e.max(), float(gold["ref_ia_per_part"])), flush=True)
    for ru in (6.0, 5.5, 5.0, 4.75, 4.5, 4.0):
        cfg, eps = case_config(pkg, wiring, ng, walk_mode=pkg.WALK_GROUP, group_reach=ru)
        eng = pkg.Engine(cfg); eng.set_particles(pos, mass, typ, old_acc=gold["old_acc"]); eng.set_opening(0.0, 0.005)
        eng.compute_accelerations(True)
        ag, _, cost, pmg = eng.get_accel(want_pm=True)
        e = np.linalg.norm((ag + pmg)[idx] - truth, axis=1) / np.linalg.norm(truth, axis=1)
        print("  group reach %.2f: rms %.3e max %.2e ia %.1f" % (ru, rms(e), e.max(), cost.mean()), flush=True)
        eng.close()
