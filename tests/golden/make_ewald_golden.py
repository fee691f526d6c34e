#!/usr/bin/env python3
"""Generate tests/golden/ewald_truth_<wiring>.npz: Ewald-summed accelerations (tests/ewald.py, written
from the textbook formulas) for a sample of targets of the seeded TreePM parity boxes, plus the oracle's
(= reference algorithm's) TreePM result on the same targets.  CPU only; run from the repo root:
    python tests/golden/make_ewald_golden.py
The GPU accuracy tests load these instead of spending minutes of GPU-box time in numpy."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
from ewald import ewald_direct  # noqa: E402

CASES = {"newton": 1, "c4": 2}
N, L, PMGRID, SEED, STRIDE = 40000, 1e4, 32, 21, 125


def case_config(pkg, wiring, ng, **kw):
    eps = L / (40 * N ** (1 / 3))
    return pkg.make_config(n_gravs=ng, periodic=1, pmgrid=PMGRID, box_size=L, G=43007.1, theta=0.5, softening=[eps] * 6,
                           type_to_grav=pkg.ic.default_type_to_grav(ng), wiring=wiring, **kw), eps


def main():
    pkg, O = ge.load_package(), ge.load_oracle()
    for wiring, ng in CASES.items():
        pos, mass, typ = pkg.ic.uniform_box(N, box=L, n_gravs=ng, seed=SEED)
        cfg, eps = case_config(pkg, wiring, ng)
        tab, _ = O.shortrange_table(cfg)
        pm = O.pm_periodic(cfg, pos, mass, typ)
        T = O.Tree(cfg, pos, mass, typ)
        a, _ = T.walk(table=tab)
        _, old = O.finish(cfg, a, pm)
        cfg.err_tol_theta = 0.0
        a2, n2 = T.walk(old_acc=old, table=tab)
        a2, _ = O.finish(cfg, a2, pm)
        idx = np.arange(0, N, STRIDE)
        species = np.array(pkg.ic.default_type_to_grav(ng))[typ]
        law = [[cfg.law_accel[i][j] for j in range(ng)] for i in range(ng)]
        truth = ewald_direct(pos, mass, species, idx, L, cfg.G, law, cfg.yukawa_imass / L, 2.8 * eps)
        ref_total = (a2 + pm)[idx]
        e = np.linalg.norm(ref_total - truth, axis=1) / np.linalg.norm(truth, axis=1)
        print(wiring, "reference walk vs Ewald: rms %.3e max %.3e ia/part %.1f" % (np.sqrt(np.mean(e ** 2)), e.max(), n2.mean()))
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ewald_truth_%s.npz" % wiring), idx=idx, truth=truth,
                            ref_total=ref_total, old_acc=old, ref_ia_per_part=n2.mean(),
                            meta=np.array([N, L, PMGRID, SEED, STRIDE, ng]))


if __name__ == "__main__":
    main()
