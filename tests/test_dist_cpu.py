"""world_size-2 rehearsal of the N>1 path on CPU (gloo): every rank holds the full particle set, owns one
contiguous Peano segment of TARGETS, and no data-path collective is needed; merging the shards must give
the single-task result exactly (the reference's invariant, domain.c:18-21).  The force engine itself
needs a GPU, so the per-shard walk here is the oracle's; the GPU twin of this test is
test_gpu_parity.py::test_pm_persists_between_pm_steps_and_multi_shard."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg, O = ge.load_package(), ge.load_oracle()
    n = 6000
    pos, mass, typ = pkg.ic.plummer_sphere(n, seed=9)
    cfg = pkg.make_config(n_gravs=1, G=1.0, theta=0.5, softening=[0.01] * 6)
    dom = O.domain_extent(pos)
    order = np.argsort(O.keys(pos, dom), kind="stable")            # Peano order of the whole set, same on every rank
    first, count = pkg.shard_range(n, rank, world)
    mine = order[first:first + count].astype(np.int32)
    T = O.Tree(cfg, pos, mass, typ, dom)
    acc, nint = T.walk(idx=mine, nthreads=1)
    # assemble: each rank contributes its rows; all_reduce(sum) of disjoint rows == gather
    full = torch.zeros((n, 3), dtype=torch.float64)
    full[torch.from_numpy(mine.astype(np.int64))] = torch.from_numpy(acc)
    dist.all_reduce(full)
    # the timing protocol of bench.py: barrier, max over ranks
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        ref, _ = T.walk(nthreads=1)
        np.save(os.path.join(out_dir, "ok.npy"), np.array([float(np.array_equal(full.numpy(), ref)), t.item(), count]))
    dist.destroy_process_group()


def test_two_rank_sharding_reproduces_single_task(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ok, tmax, count = np.load(os.path.join(str(tmp_path), "ok.npy"))
    assert ok == 1.0
    assert abs(tmax - 0.2) < 1e-12


@pytest.mark.parametrize("n,ws", [(1, 1), (63, 2), (64, 2), (1000, 3), (1 << 20, 8), (12345, 8)])
def test_shards_partition_the_peano_order(pkg, n, ws):
    covered = 0
    for r in range(ws):
        first, count = pkg.shard_range(n, r, ws)
        assert first == covered and count >= 0
        assert first % 64 == 0
        covered += count
    assert covered == n
