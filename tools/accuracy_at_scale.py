#!/usr/bin/env python3
"""GPU: force accuracy of the production path AT THE BENCH SIZE.  Runs the C4 workload of bench.py (default 2^26
particles, N_GRAVS=2, c4 wiring, PMGRID=512, relative criterion) and compares tree+PM accelerations of a random sample
of targets with the periodic direct sum (all sources, nearest image + lattice-sum correction tables -- the reference's
gravity_forcetest() path, forcetree.c:3428-3548) computed on the same GPU.  Writes profiles/<tag>_accuracy.json.

usage: accuracy_at_scale.py [--log2n 26] [--samples 256] [--tag r01]
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
    ap.add_argument("--log2n", type=int, default=26)
    ap.add_argument("--samples", type=int, default=256)
    ap.add_argument("--tag", default="r01")
    ap.add_argument("--ngravs", type=int, default=2)
    ap.add_argument("--wiring", default="c4")
    ap.add_argument("--pmgrid", type=int, default=0)
    args = ap.parse_args()
    import torch
    pkg = ge.load_package()
    n, L, ng = 1 << args.log2n, 1.0, args.ngravs
    pmgrid = args.pmgrid
    if not pmgrid:
        pmgrid = 16
        while (pmgrid * 2) ** 3 <= 2 * n:
            pmgrid *= 2
    eps = L / (40 * n ** (1 / 3))
    cfg = pkg.make_config(n_gravs=ng, periodic=1, pmgrid=pmgrid, box_size=L, G=1.0, theta=0.5, err_tol_force_acc=0.005,
                          softening=[eps] * 6, type_to_grav=pkg.ic.default_type_to_grav(ng), wiring=args.wiring,
                          walk_mode=pkg.WALK_GROUP)
    pos, mass, ptype = bench.make_box(pkg, n, L, ng, 12345)
    dev = torch.device("cuda", 0)
    d_pos, d_mass, d_type = torch.from_numpy(pos).to(dev), torch.from_numpy(mass).to(dev), torch.from_numpy(ptype).to(dev)
    d_old = torch.zeros(n, dtype=torch.float64, device=dev)
    del pos, mass, ptype
    eng = pkg.Engine(cfg)
    eng.set_particles_device(n, d_pos.data_ptr(), d_mass.data_ptr(), d_type.data_ptr())
    eng.compute_accelerations(pm_step=True)           # theta pass
    eng.get_old_acc_device(d_old.data_ptr())
    eng.set_old_acc_device(d_old.data_ptr())
    eng.set_opening(0.0, 0.005)
    eng.compute_accelerations(pm_step=True)           # relative criterion: the timed configuration of bench.py
    print("tree+PM done", flush=True)
    acc, _, cost, pm = eng.get_accel(want_pm=True)
    idx = np.sort(np.random.default_rng(7).choice(n, args.samples, replace=False)).astype(np.int32)
    tot = (acc + pm)[idx]
    ia = float(cost.mean())
    del acc, pm, cost
    t0 = time.time()
    truth = np.zeros((len(idx), 3))
    chunk = 32                                         # progress lines for the runner's watchdog
    for s in range(0, len(idx), chunk):
        truth[s:s + chunk] = eng.direct_sum(idx[s:s + chunk])
        print("direct sum %d/%d (%.0f s)" % (min(len(idx), s + chunk), len(idx), time.time() - t0), flush=True)
    e = np.linalg.norm(tot - truth, axis=1) / np.linalg.norm(truth, axis=1)
    out = {"workload": "%d particles, N_GRAVS=%d (%s wiring), PMGRID=%d, ErrTolForceAcc=0.005, group walk" % (n, ng, args.wiring, pmgrid),
           "samples": int(len(idx)), "truth": "periodic direct sum on the GPU (nearest image + lattice correction tables)",
           "rel_err_rms": float(np.sqrt(np.mean(e ** 2))), "rel_err_median": float(np.median(e)), "rel_err_p99": float(np.percentile(e, 99)),
           "rel_err_max": float(e.max()), "interactions_per_particle": ia,
           "reference_band": "reference TreePM walk vs Ewald at the same ErrTolForceAcc: rms 6.5e-3 (Newton+Yukawa), 9.1e-3 (Newton); SURVEY.md 6",
           "direct_sum_seconds": time.time() - t0}
    print(json.dumps(out), flush=True)
    # gpurun only brings gpurun_out/ back from the GPU box; the file is then copied into profiles/
    for d in ("gpurun_out", "profiles"):
        os.makedirs(os.path.join(ROOT, d), exist_ok=True)
        with open(os.path.join(ROOT, d, "%s_accuracy_2p%d_ng%d.json" % (args.tag, args.log2n, ng)), "w") as f:
            json.dump(out, f, indent=1)
    eng.close()


if __name__ == "__main__":
    main()
