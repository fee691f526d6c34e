#!/usr/bin/env python3
"""Interactions per particle and walk time of the production walk on a CLUSTERED TreePM box (60 % of the particles in one clump) for
its default (the unit size follows the pairs-per-target ratio of the last walk) and for traversal units fixed at 4, 2 and 1 groups,
next to the reference walk: what the shared decisions cost outside the uniform regime.
  python tools/clustered_probe.py [log2n]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge


def main():
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    pkg = ge.load_package()
    n, L, ng = 1 << log2n, 1.0, 2
    rng = np.random.default_rng(77)
    pos = rng.random((n, 3))
    k = int(0.6 * n)
    pos[:k] = np.mod(0.5 + 0.05 * rng.standard_normal((k, 3)), 1.0)
    mass = np.full(n, 1.0 / n)
    typ = (1 + rng.integers(0, ng, n)).astype(np.int32)
    pmgrid = 16
    while (pmgrid * 2) ** 3 <= 2 * n:
        pmgrid *= 2
    eps = L / (40 * n ** (1 / 3))
    out = {"particles": n, "pmgrid": pmgrid, "clump": "60 % of the particles in a Gaussian of sigma 0.05 L"}
    for name, mode, tune in (("reference_walk", pkg.WALK_STRICT, {}), ("production_default", pkg.WALK_GROUP, {}),
                             ("production_sg4", pkg.WALK_GROUP, {"walk_sg": 4}), ("production_sg2", pkg.WALK_GROUP, {"walk_sg": 2}),
                             ("production_sg1", pkg.WALK_GROUP, {"walk_sg": 1}),
                             ("production_sg1_spread2", pkg.WALK_GROUP, {"walk_sg": 1, "walk_spread": 2}),
                             ("production_sg1_spread4", pkg.WALK_GROUP, {"walk_sg": 1, "walk_spread": 4}),
                             ("production_sg2_spread2", pkg.WALK_GROUP, {"walk_sg": 2, "walk_spread": 2}),
                             ("production_sg4_spread2", pkg.WALK_GROUP, {"walk_sg": 4, "walk_spread": 2}),
                             ("production_sg4_spread4", pkg.WALK_GROUP, {"walk_sg": 4, "walk_spread": 4})):
        cfg = pkg.make_config(n_gravs=ng, periodic=1, pmgrid=pmgrid, box_size=L, G=1.0, theta=0.5, softening=[eps] * 6,
                              type_to_grav=pkg.ic.default_type_to_grav(ng), wiring="c4", walk_mode=mode)
        eng = pkg.Engine(cfg)
        if tune:
            eng.set_tuning(**tune)
        eng.set_particles(pos, mass, typ)
        eng.compute_accelerations(pm_step=True)
        _, old, _ = eng.get_accel()
        eng.set_old_acc(old)
        eng.set_opening(0.0, 0.005)
        eng.gravity_tree()
        eng.gravity_tree()
        st = eng.stats()
        _, _, cost = eng.get_accel()
        out[name] = {"ia_per_particle": float(cost.mean()), "ia_max": float(cost.max()), "treewalk_ms": 1e3 * st.t_treewalk}
        eng.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
