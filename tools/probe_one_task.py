#!/usr/bin/env python3
"""Stage times of the multi-task step on ONE task that has the GPU to itself (the rehearsals with several tasks on one GPU time
kernels that compete for it): 2^23 particles and PMGRID 256 -- the particle count and the number of mesh cells ONE of 8 tasks
holds at C4 (2^26 particles, PMGRID 512) -- through the production backend (torch.distributed "nccl" = RCCL, world size 1).
Imports and exchange payloads are absent (no other task), everything else of the choreography runs.
  python tools/probe_one_task.py [log2n] [pmgrid] [nokept]"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
import __graft_entry__ as ge
import bench


def main():
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 23
    pmgrid = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 400))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    pkg = ge.load_package()
    dd = importlib.import_module("ngravs_amd.distributed")
    n, L = 1 << log2n, 1.0
    eps = L / (40 * n ** (1 / 3))
    cfg = pkg.make_config(n_gravs=2, periodic=1, pmgrid=pmgrid, box_size=L, G=1.0, theta=0.5, err_tol_force_acc=0.005,
                          softening=[eps] * 6, type_to_grav=pkg.ic.default_type_to_grav(2), wiring="c4", walk_mode=pkg.WALK_GROUP)
    pos, mass, ptype = bench.make_box(pkg, n, L, 2, 12345)
    eng = dd.DistributedEngine(cfg)
    eng.set_particles(pos, mass, ptype, ids=np.arange(n))
    eng.compute_accelerations(pm_step=True)
    nl = eng.num_local()
    tmp = torch.zeros(nl, dtype=torch.float64, device="cuda")
    eng.get_old_acc_device(tmp.data_ptr())
    eng._check(pkg.lib().ngravs_set_old_acc(eng._h, tmp.data_ptr(), 8, 1), "ngravs_set_old_acc")
    eng.set_opening(0.0, 0.005)
    eng.compute_accelerations(pm_step=True)
    eng.reset_wall()
    steps = 3
    dd_s, pm_s = np.zeros(8), np.zeros(13)
    for _ in range(steps):
        eng.compute_accelerations(pm_step=True)
        dd_s += np.array(list(eng.info.seconds))
        pm_s += np.array(eng.pm_seconds())
    st = eng.stats()
    w = eng.wall
    names = bench.DD_STAGES
    calls, secs, byt = eng.comm.stats() if hasattr(eng.comm, "stats") else (w["collective_calls"], 0.0, 0.0)
    out = {"particles": n, "pmgrid": pmgrid, "backend": "%s, world size %d" % (eng.backend, eng.comm.world_reported),
           "top_tree": {"nodes": int(eng.info.n_topnodes), "leaves": int(eng.info.n_topleaves), "counting_rounds": int(eng.info.toptree_rounds)},
           "decomposition_collectives_last_step": int(eng.info.collectives),
           "communicator_totals": {"calls": calls, "seconds_inside": secs, "bytes": byt, "mean_latency_ms": 1e3 * secs / max(1, calls)},
           "wall_ms_per_step": {k[:-2]: 1e3 * v / steps for k, v in w.items() if k.endswith("_s")},
           "collective_calls_per_step": w["collective_calls"] / steps,
           "decomposition_stage_ms": {k: 1e3 * v / steps for k, v in zip(names, dd_s)},
           "pm_stage_ms": {"deposit+boxes": 1e3 * pm_s[0] / steps, "pack": list(1e3 * pm_s[1::3] / steps),
                           "alltoallv": list(1e3 * pm_s[2::3] / steps), "unpack": list(1e3 * pm_s[3::3] / steps)},
           "device_phases_ms_last_step": {"domain+peano": st.t_domain + st.t_peano, "pm": st.t_pm, "treebuild": st.t_treebuild,
                                          "treewalk": st.t_treewalk}}
    # steps that KEEP the decomposition (TreeDomainUpdateFrequency > 0): the own rows drift, ngravs_host_kept_step does the rest
    import ctypes as C
    import time
    if len(sys.argv) > 3 and sys.argv[3] == "nokept":   # (tools/trace_one_task.sh: the trace ends with a step that decomposes)
        print(json.dumps(out))
        eng.close()
        dist.destroy_process_group()
        return
    ids = eng.local_ids()
    pl, ml, tl = np.asarray(pos)[ids].copy(), np.asarray(mass)[ids].copy(), np.asarray(ptype)[ids].copy()
    rng = np.random.default_rng(3)
    kept = {"host_ms": [], "stage_ms": [], "walk_ms": []}
    for _ in range(3):
        pl += 2e-4 * L * rng.standard_normal(pl.shape)
        eng.n = eng.num_local()
        eng.update_particles(pl, ml, tl)
        eng._check(pkg.lib().ngravs_set_old_acc(eng._h, tmp.data_ptr(), 8, 1), "ngravs_set_old_acc")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng._host(eng._L.ngravs_host_kept_step(eng._h, C.byref(eng.comm.c), C.byref(eng.info)), "ngravs_host_kept_step")
        t1 = time.perf_counter()
        eng.gravity_tree()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        kept["host_ms"].append(1e3 * (t1 - t0))
        kept["walk_ms"].append(1e3 * (t2 - t1))
        kept["stage_ms"].append({k: 1e3 * v for k, v in zip(names, list(eng.info.seconds))})
    kept["collectives"] = int(eng.info.collectives)
    kept["unopened"] = int(eng.walk_unopened())
    out["kept_step"] = kept
    print(json.dumps(out))
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
