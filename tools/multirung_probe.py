#!/usr/bin/env python3
"""GPU: cost of a force computation for small active sets (individual timesteps) and of the dynamic tree update, C4-like
box.  Prints ms per call of gravity_tree() for several active fractions with and without target compaction, and of
ngravs_force_update_tree() against domain_Decomposition()+force_treebuild().

usage: multirung_probe.py [--log2n 24]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge   # noqa: E402
import bench                   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=24)
    args = ap.parse_args()
    import torch
    pkg = ge.load_package()
    n, L, ng = 1 << args.log2n, 1.0, 2
    pmgrid = 16
    while (pmgrid * 2) ** 3 <= 2 * n:
        pmgrid *= 2
    eps = L / (40 * n ** (1 / 3))
    cfg = pkg.make_config(n_gravs=ng, periodic=1, pmgrid=pmgrid, box_size=L, G=1.0, theta=0.5, err_tol_force_acc=0.005,
                          softening=[eps] * 6, type_to_grav=pkg.ic.default_type_to_grav(ng), wiring="c4",
                          walk_mode=pkg.WALK_GROUP)
    pos, mass, ptype = bench.make_box(pkg, n, L, ng, 12345)
    dev = torch.device("cuda", 0)
    d_pos, d_mass, d_type = torch.from_numpy(pos).to(dev), torch.from_numpy(mass).to(dev), torch.from_numpy(ptype).to(dev)
    d_old = torch.zeros(n, dtype=torch.float64, device=dev)
    eng = pkg.Engine(cfg)
    eng.set_particles_device(n, d_pos.data_ptr(), d_mass.data_ptr(), d_type.data_ptr())
    eng.compute_accelerations(pm_step=True)
    eng.get_old_acc_device(d_old.data_ptr())
    eng.set_opening(0.0, 0.005)
    out = {"particles": n, "pmgrid": pmgrid}

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    rng = np.random.default_rng(5)
    for frac in (1.0, 0.3, 0.1, 0.01, 0.001):
        act = (rng.uniform(size=n) < frac).astype(np.uint8) if frac < 1.0 else np.ones(n, dtype=np.uint8)
        d_act = torch.from_numpy(act).to(dev)
        for compact in ("1", "0"):
            eng.set_tuning(walk_compact=int(compact))
            eng.set_particles_device(n, d_pos.data_ptr(), d_mass.data_ptr(), d_type.data_ptr(), old_acc_ptr=d_old.data_ptr(),
                                     active_ptr=d_act.data_ptr())
            eng.domain_Decomposition()
            eng.force_treebuild()
            ms = timed(eng.gravity_tree)
            out["walk_ms_active_%g_compact_%s" % (frac, compact)] = ms
            print("active %.3f compact %s: gravity_tree %.2f ms (%.1f ns per active target)" %
                  (frac, compact, ms, ms * 1e6 / max(1, int(act.sum()))), flush=True)
        eng.set_tuning(walk_compact=1)

    def rebuild():
        eng.domain_Decomposition()
        eng.force_treebuild()
    out["rebuild_ms"] = timed(rebuild)
    out["refit_ms"] = timed(eng.force_update_tree)
    print("decomposition + build %.2f ms | refit (force_update_tree) %.2f ms" % (out["rebuild_ms"], out["refit_ms"]), flush=True)
    print(json.dumps(out), flush=True)
    for d in ("gpurun_out", "profiles"):
        os.makedirs(os.path.join(ROOT, d), exist_ok=True)
        with open(os.path.join(ROOT, d, "r01_multirung_2p%d.json" % args.log2n), "w") as f:
            json.dump(out, f, indent=1)
    eng.close()


if __name__ == "__main__":
    main()
