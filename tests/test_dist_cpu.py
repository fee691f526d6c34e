"""CPU (gloo, world size 2-3, no GPU): the pure-C pieces of the multi-task decomposition (host/ngravs_host.c) driven the way
ngravs_host_domain_owners / ngravs_host_domain_halo drive them -- top-tree rounds on all-reduced leaf counts
(ngravs_host_toptree_adapt), the cut (ngravs_host_split), and the import decision (ngravs_host_import_request) -- checked
against the oracle's walk: every part of the tree the reference walk of one of a task's targets enters must be on that task.

Reference: domain_determineTopTree / domain_topsplit (domain.c:933-1138), domain_findSplit / domain_shiftSplit (:347-544),
the export decision of force_treeevaluate (forcetree.c:1424-1434: a target that opens a pseudo particle is exported; here the
leaf is imported instead)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _case(pkg, name):
    n = 12000
    if name.startswith("plummer"):
        pos, mass, typ = pkg.ic.plummer_sphere(n, seed=17)
        typ = (1 + (np.arange(n) % 2)).astype(np.int32)
        cfg = pkg.make_config(n_gravs=2, G=1.0, theta=0.5, softening=[0.01, 0.01, 0.03, 0.01, 0.01, 0.01],
                              type_to_grav=pkg.ic.default_type_to_grav(2), wiring="newton")
    else:
        L = 1.0
        pos, mass, typ = pkg.ic.uniform_box(n, box=L, n_gravs=2, seed=23)
        eps = L / (40 * n ** (1 / 3))
        cfg = pkg.make_config(n_gravs=2, periodic=1, pmgrid=32, box_size=L, G=1.0, theta=0.5, softening=[eps] * 6,
                              type_to_grav=pkg.ic.default_type_to_grav(2), wiring="c4")
        cfg.asmth = 1.25 * L / 32          # what ngravs_get_config() reports (pm_periodic.c:59-60)
        cfg.rcut = 4.5 * cfg.asmth
    if name.endswith("rel"):
        cfg.err_tol_theta = 0.0            # relative criterion (gravtree.c:334-335)
    return pos, mass, typ, cfg


def _worker(rank, world, port, out_dir, name):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    from test_abi import _tree_struct, leaf_counts
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg, O = ge.load_package(), ge.load_oracle()
    L = pkg.lib()
    pos, mass, typ, cfg = _case(pkg, name)
    n, ng = len(pos), cfg.n_gravs
    cw = 7 + 4 * ng
    dom = O.domain_extent(pos)
    key21 = O.keys(pos, dom).astype(np.int64) << 9
    mine0 = np.arange(rank, n, world)                     # what this task holds before the decomposition
    # a steady-state OldAcc for the relative criterion: the reference walk with theta
    T = O.Tree(cfg, pos, mass, typ, dom)
    tab = O.shortrange_table(cfg)[0] if cfg.pmgrid else None
    cfg_theta = pkg.make_config(n_gravs=1)
    C.memmove(C.byref(cfg_theta), C.byref(cfg), C.sizeof(cfg))
    cfg_theta.err_tol_theta = 0.5
    a0, _ = T.walk(table=tab, cfg=cfg_theta, nthreads=4)
    old = np.linalg.norm(a0, axis=1)
    # ---- domain_determineTopTree: rounds of the rule on ALL-REDUCED leaf counts
    TopTree = _tree_struct()
    t = TopTree()
    assert L.ngravs_host_toptree_init(C.byref(t), 2) == 0
    thresh = n / (20.0 * world)                           # TotNumPart / (TOPNODEFACTOR * NTask), domain.c:1127
    rounds = 0
    while True:
        cnt = torch.from_numpy(leaf_counts(t, key21[mine0]))
        dist.all_reduce(cnt)
        cnt = cnt.numpy()
        nxt = TopTree()
        unknown = L.ngravs_host_toptree_adapt(C.byref(t), cnt.ctypes.data, thresh, 18, C.byref(nxt))
        assert unknown >= 0
        rounds += 1
        if unknown == 0 and nxt.nnode in (0, t.nnode):
            L.ngravs_host_toptree_free(C.byref(nxt))
            break
        L.ngravs_host_toptree_free(C.byref(t))
        t = nxt
    nn, nl = t.nnode, t.nleaf
    child = np.ctypeslib.as_array(t.child, (nn,)).copy()
    level = np.ctypeslib.as_array(t.level, (nn,)).copy()
    leaf = np.ctypeslib.as_array(t.leaf, (nn,)).copy()
    node_of_leaf = np.ctypeslib.as_array(t.node_of_leaf, (nl,)).copy()
    assert cnt.sum() == n and cnt.max() <= thresh
    # ---- leaf of every particle, per-leaf sums of the own particles, all-reduced (ngravs_dd_leaf_sums' layout)
    start = np.zeros(nn, dtype=np.int64)
    for i in range(nn):
        if child[i] >= 0:
            for k in range(8):
                start[child[i] + k] = int(start[i]) + (k << (3 * (21 - int(level[i]) - 1)))
    lo = start[node_of_leaf].astype(np.uint64)
    leaf_of = np.searchsorted(lo, key21.astype(np.uint64), side="right") - 1
    sums = np.zeros((nl, cw))
    t2g = np.array([cfg.type_to_grav[k] for k in range(6)])
    for i in mine0:
        q = sums[leaf_of[i]]
        q[0] += 1.0                                        # work: 1 + GravCost with GravCost = 0
        q[1 + typ[i]] += 1.0
        g = t2g[typ[i]]
        q[7 + 4 * g] += mass[i]
        q[8 + 4 * g: 11 + 4 * g] += mass[i] * pos[i]
    ts = torch.from_numpy(sums)
    dist.all_reduce(ts)
    sums = ts.numpy()
    lcount = sums[:, 1:7].sum(axis=1)
    assert np.array_equal(lcount, cnt)
    # ---- the cut
    owner = np.zeros(nl, dtype=np.int32)
    assert L.ngravs_host_split(lcount.ctypes.data, sums[:, 0].copy().ctypes.data, nl, world, 1.5 * n / world, owner.ctypes.data) == 0
    # ---- sums of every top node
    node_sums = np.zeros((nn, cw))
    for i in range(nn - 1, -1, -1):
        if child[i] < 0:
            node_sums[i] = sums[leaf[i]]
            node_sums[i, 0] = lcount[leaf[i]]
        else:
            node_sums[i] = node_sums[child[i]: child[i] + 8].sum(axis=0)
    # ---- the import decision of THIS task
    targets = np.flatnonzero(owner[leaf_of] == rank).astype(np.int32)
    bounds = np.array([(cfg.err_tol_force_acc * old[targets]).min(), min(cfg.force_softening[k] for k in set(typ[targets].tolist()))])
    need = np.zeros(nl, dtype=np.uint8)
    dom_c = np.ascontiguousarray(dom, dtype=np.float64)
    rc = L.ngravs_host_import_request(C.byref(cfg), dom_c.ctypes.data, C.byref(t), node_sums.ctypes.data, owner.ctypes.data, rank,
                                      bounds.ctypes.data, need.ctypes.data)
    assert rc == 0
    # a decomposition that will be kept (ngravs_host_import_request_margin: own boxes grown by the drift allowance) asks for at least
    # as much, and margin 0 is the plain decision
    L.ngravs_host_import_request_margin.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_double,
                                                    C.c_void_p]
    need0, needm = np.zeros(nl, dtype=np.uint8), np.zeros(nl, dtype=np.uint8)
    for mg, out in ((0.0, need0), (0.01 * dom[6], needm)):
        assert L.ngravs_host_import_request_margin(C.byref(cfg), dom_c.ctypes.data, C.byref(t), node_sums.ctypes.data, owner.ctypes.data, rank,
                                                   bounds.ctypes.data, mg, out.ctypes.data) == 0
    assert np.array_equal(need0, need) and np.all(needm >= need) and needm.sum() >= need.sum()
    # ---- what the reference walk of these targets really enters
    reach = T.walk_reach(targets, old_acc=old, table=tab)
    leaf_len = dom[6] / (1 << level[node_of_leaf]).astype(np.float64)
    entered = reach < leaf_len[leaf_of] * (1 - 1e-9)      # the particle was used below the size of its top leaf
    must = np.zeros(nl, dtype=bool)
    must[np.unique(leaf_of[entered])] = True
    must &= owner != rank
    have = need.astype(bool)
    np.savez(os.path.join(out_dir, "n%d.npz" % rank), missing=np.flatnonzero(must & ~have), must=must.sum(), have=have.sum(),
             nleaf=nl, rounds=rounds, imported=lcount[have].sum(), own=len(targets), owner=owner, tree=child)
    T.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["plummer", "plummer_rel", "treepm", "treepm_rel"])
def test_import_decision_covers_the_reference_walk(pkg, have_lib, O, tmp_path, name):
    """For every task: the set of foreign top leaves ngravs_host_import_request asks for is a superset of the leaves some particle
    of which the reference walk (oracle) of one of the task's targets uses individually or through a node smaller than the leaf.
    Tree-only Plummer sphere (two softening lengths) and periodic TreePM box, Barnes-Hut and relative criterion; every task builds
    the same tree and the same cut."""
    import torch.multiprocessing as mp
    world = 3 if name.startswith("plummer") else 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path), name), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), "n%d.npz" % r)) for r in range(world)]
    for r, d in enumerate(res):
        print("%s task %d: top tree %d leaves (%d rounds); %d own particles, %d imported; %d leaves entered by the walk, %d requested" %
              (name, r, d["nleaf"], d["rounds"], d["own"], d["imported"], d["must"], d["have"]))
        assert len(d["missing"]) == 0, "leaves the walk enters but the task did not ask for: %s" % d["missing"][:10]
        assert d["have"] >= d["must"]
        assert np.array_equal(d["owner"], res[0]["owner"]) and np.array_equal(d["tree"], res[0]["tree"])
    if name.startswith("plummer"):
        # the adaptive leaves keep the import of a centrally concentrated set well below "everything"
        imp = sum(float(d["imported"]) for d in res)
        own = sum(float(d["own"]) for d in res)
        print("%s: imported / own = %.2f" % (name, imp / own))


@pytest.mark.parametrize("n,ws", [(1, 1), (63, 2), (64, 2), (1000, 3), (1 << 20, 8), (12345, 8)])
def test_shards_partition_the_peano_order(pkg, n, ws):
    covered = 0
    for r in range(ws):
        first, count = pkg.shard_range(n, r, ws)
        assert first == covered and count >= 0
        assert first % 64 == 0
        covered += count
    assert covered == n


def _selftest_worker(rank, world, port, out_dir, fail_rank, fail_stage):
    """a host-buffer communicator over gloo (Python callbacks in a struct ngravs_comm) through ngravs_host_comm_selftest"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import time
    import importlib
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = ge.load_package()
    dd = importlib.import_module(pkg.__name__ + ".distributed")
    L = pkg.lib()
    ncalls = [0]

    def view(ptr, nbytes):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(int(nbytes),)) if nbytes > 0 else np.zeros(0, np.uint8)

    def allreduce(user, buf, count, dtype, op):
        ncalls[0] += 1
        h = view(buf, 8 * count).view(np.float64 if dtype == 0 else np.int64)
        t = torch.from_numpy(h.copy())
        dist.all_reduce(t, op={0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MIN, 2: dist.ReduceOp.MAX}[op])
        h[:] = t.numpy()
        return 0

    def allgather(user, send, recv, nbytes):
        ncalls[0] += 1
        t = torch.from_numpy(view(send, nbytes).copy())
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        view(recv, world * nbytes)[:] = torch.cat(outs).numpy()
        return 0

    def alltoallv(user, send, sbytes, sdispl, recv, rbytes, rdispl):
        ncalls[0] += 1
        sb, rb = [int(sbytes[r]) for r in range(world)], [int(rbytes[r]) for r in range(world)]
        pad = max(1, max(sum(sb), 1))
        mx = torch.tensor([pad], dtype=torch.int64)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        buf = torch.zeros(int(mx.item()) + 8 * world, dtype=torch.uint8)
        hdr = torch.tensor(sb, dtype=torch.int64)
        buf[:8 * world] = torch.from_numpy(hdr.numpy().view(np.uint8).copy())
        buf[8 * world: 8 * world + sum(sb)] = torch.from_numpy(view(send, sum(sb)).copy())
        outs = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(outs, buf)
        out = view(recv, sum(rb))
        at = 0
        for r in range(world):
            cnt = outs[r][:8 * world].numpy().view(np.int64)
            off = int(cnt[:rank].sum())
            out[at: at + rb[r]] = outs[r][8 * world + off: 8 * world + off + int(cnt[rank])].numpy()
            at += rb[r]
        return 0

    cbs = (dd._ALLREDUCE(allreduce), dd._ALLGATHER(allgather), dd._ALLTOALLV(alltoallv))
    cm = dd.Comm(rank, world, 0, 0, None, *cbs, C.cast(None, dd._ALLREDUCE))
    L.ngravs_host_comm_selftest.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_int]
    why = C.create_string_buffer(200)
    t0 = time.time()
    st = L.ngravs_host_comm_selftest(None, C.byref(cm), fail_stage if rank == fail_rank else 0, why, 200)
    np.savez(os.path.join(out_dir, "st%d.npz" % rank), status=st, seconds=time.time() - t0, calls=ncalls[0], why=why.value.decode())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_stage", [0, 1, 2, 3])
def test_comm_selftest_is_collective_safe(pkg, have_lib, tmp_path, fail_stage):
    """ngravs_host_comm_selftest (include/ngravs_comm_selftest.h; the same code is ngravs_rccl_selftest): three tasks over gloo, task 1
    is made to find a wrong answer in stage `fail_stage`.  Every task must still run every collective (equal call counts: nobody
    leaves early and lets the others wait), all three must return the SAME status -- the failing stage's bit, or 0 -- within seconds."""
    import torch.multiprocessing as mp
    world = 3
    port = 31500 + (os.getpid() % 2000) + fail_stage
    mp.spawn(_selftest_worker, args=(world, port, str(tmp_path), 1, fail_stage), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), "st%d.npz" % r)) for r in range(world)]
    want = 0 if fail_stage == 0 else 1 << (fail_stage - 1)
    for r, d in enumerate(res):
        print("stage %d: task %d status %d after %.2f s, %d collective callbacks, says %r" % (fail_stage, r, d["status"], d["seconds"], d["calls"], str(d["why"])))
        assert int(d["status"]) == want
        assert int(d["calls"]) == int(res[0]["calls"]) and float(d["seconds"]) < 20
    if fail_stage:
        assert "self test" in str(res[1]["why"]) and "another task" in str(res[0]["why"])
