#!/usr/bin/env python3
"""debug: 3 tasks on one GPU (gloo), plummer strict; python tools/debug_dist.py LEAF_MAX"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

def worker(rank, world, port, out, leaf_max):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import importlib, torch.distributed as dist
    import __graft_entry__ as ge
    from test_gpu_dist import _strict_case
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = ge.load_package(); dd = importlib.import_module("ngravs_amd.distributed")
    pos, mass, typ, old, cfg = _strict_case(pkg, "plummer")
    n = len(pos); mine = np.arange(rank, n, world)
    eng = dd.DistributedEngine(cfg, leaf_max=leaf_max)
    eng.set_particles(pos[mine], mass[mine], typ[mine], old_acc=old[mine], ids=mine)
    eng.compute_accelerations(pm_step=False)
    acc, oa, cost = eng.get_accel()[:3]
    st = eng.stats()
    np.savez(os.path.join(out, "s%d.npz" % rank), ids=eng.local_ids(), acc=acc, cost=cost, info=np.array([eng.info.n_topnodes, eng.info.n_topleaves, eng.info.toptree_rounds, eng.info.n_halo, eng.info.n_local, st.n_nodes]))
    eng.close(); dist.destroy_process_group()

if __name__ == "__main__":
    import tempfile, torch.multiprocessing as mp
    import __graft_entry__ as ge
    from test_gpu_dist import _strict_case
    pkg = ge.load_package()
    pos, mass, typ, old, cfg = _strict_case(pkg, "plummer")
    n = len(pos)
    eng = pkg.Engine(cfg); eng.set_particles(pos, mass, typ, old_acc=old); eng.compute_accelerations(pm_step=False)
    a1, _, c1 = eng.get_accel(); print("single nodes", eng.stats().n_nodes); eng.close()
    for world in (2, 3):
        for lm in [float(x) for x in sys.argv[1:]] or [1e9]:
            out = tempfile.mkdtemp()
            mp.spawn(worker, args=(world, 29800 + os.getpid() % 1000, out, lm), nprocs=world, join=True)
            acc = np.zeros((n, 3)); cost = np.zeros(n); owner = np.zeros(n, int)
            for r in range(world):
                d = np.load(os.path.join(out, "s%d.npz" % r)); acc[d["ids"]] = d["acc"]; cost[d["ids"]] = d["cost"]; owner[d["ids"]] = r
                print("world", world, "leaf_max", lm, "task", r, "topnodes/leaves/rounds/halo/local/treenodes", d["info"])
            err = np.linalg.norm(acc - a1, axis=1) / np.linalg.norm(a1, axis=1)
            bad = np.flatnonzero(cost != c1)
            print("  max err %.2e, counts differ for %d particles; by task %s; cost diff sample %s" % (err.max(), len(bad), np.bincount(owner[bad], minlength=world), (cost - c1)[bad][:8]))
