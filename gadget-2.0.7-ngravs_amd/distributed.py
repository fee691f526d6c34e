"""Multi-task (one process per GPU) host layer: Peano-Hilbert domain decomposition with particle migration and
a short-range halo, the role of domain_Decomposition() + the export/import loop of gravity_tree() in the
reference (domain.c:164-330, 554-760; gravtree.c:112-285).

The C library only packs and unpacks (include/ngravs_hip.h "multi-task domain decomposition"); the
collectives are done here with torch.distributed -- backend "nccl" (= RCCL over xGMI) on device tensors in
production, "gloo" on host copies for the CPU-side rehearsal.  Collectives per step:
  all-reduce(min/max) of the extent [6 doubles]  -- domain.c:906-907
  all-reduce(sum) of the Peano-cell histogram    -- domain_sumCost, domain.c:869-871
  all-to-all-v of migrating particles (48 B)     -- domain_exchangeParticles, domain.c:695-747
  all-to-all-v of halo particles (48 B)          -- replaces gravtree.c:195-257 (targets out, partial forces back)
  all-reduce(sum) of the density mesh            -- replaces the patch -> slab shipping of pm_periodic.c:333-427
"""
import ctypes as C

import numpy as np

from . import Engine, NgravsError, lib, peano_hilbert_key

_REC = 6          # doubles per record


class _DevArray:
    """zero-copy view of library device memory for torch (CUDA array interface)"""

    def __init__(self, ptr, shape, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def _ph_of_xyz(level):
    nc = 1 << level
    out = np.empty((nc, nc, nc), dtype=np.int64)
    for x in range(nc):
        for y in range(nc):
            for z in range(nc):
                out[x, y, z] = peano_hilbert_key(x, y, z, level)
    return out


def cut_curve(hist, world_size):
    """owner of every Peano cell: contiguous runs of cells with ~equal particle counts (domain_findSplit by count)"""
    hist = np.asarray(hist, dtype=np.float64)
    tot = hist.sum()
    if tot <= 0:
        return np.zeros(len(hist), dtype=np.int32)
    mid = np.cumsum(hist) - 0.5 * hist
    return np.minimum(world_size - 1, np.floor(mid * world_size / tot)).astype(np.int32)


class DistributedEngine(Engine):
    def __init__(self, cfg, level=None, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        if self.world > 64:
            raise NgravsError("at most 64 tasks")
        cfg.rank, cfg.world_size = 0, 1          # the library sees its working set (own + halo) as a single task
        super().__init__(cfg)
        self.level = level
        self._ph = None
        self._L = lib()
        self._L.ngravs_dd_num_local.restype = C.c_int64
        self._L.ngravs_dd_num_local.argtypes = [C.c_void_p]
        self.timings = {}

    # ---- helpers -----------------------------------------------------------------------------------------
    def _torch(self):
        import torch
        return torch

    def _dev(self):
        torch = self._torch()
        return torch.device("cuda", self.cfg.device)

    def _allreduce_host(self, arr, op):
        torch = self._torch()
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if self.backend == "nccl":
            t = t.to(self._dev())
        self.dist.all_reduce(t, op=op, group=self.group)
        return t.cpu().numpy()

    def _choose_level(self):
        if self.level is not None:
            return self.level
        # the coarsest cells that are still at least as wide as the short-range cut, at most 32^3 of them
        reach = 6.0 * 1.25 * self.cfg.box_size / self.cfg.pmgrid
        lvl = 1
        while lvl < 5 and self.cfg.box_size / (1 << (lvl + 1)) >= 1.05 * reach:
            lvl += 1
        return lvl

    def _alltoallv(self, counts, dev_ptr, nrec):
        torch, dist = self._torch(), self.dist
        ws = self.world
        send_counts = torch.tensor(list(counts), dtype=torch.int64)
        gathered = [torch.zeros(ws, dtype=torch.int64) for _ in range(ws)]
        if self.backend == "nccl":
            sc = send_counts.to(self._dev())
            gl = [g.to(self._dev()) for g in gathered]
            dist.all_gather(gl, sc, group=self.group)
            mat = torch.stack([g.cpu() for g in gl])
        else:
            dist.all_gather(gathered, send_counts, group=self.group)
            mat = torch.stack(gathered)
        recv_counts = [int(mat[r, self.rank]) for r in range(ws)]
        nrecv = sum(recv_counts)
        if nrec > 0:
            inp = torch.as_tensor(_DevArray(dev_ptr, (nrec, _REC)), device=self._dev())
        else:
            inp = torch.zeros((0, _REC), dtype=torch.float64, device=self._dev())
        out = torch.empty((nrecv, _REC), dtype=torch.float64, device=self._dev())
        in_splits = [int(c) for c in counts]
        if self.backend == "nccl":
            dist.all_to_all_single(out, inp, recv_counts, in_splits, group=self.group)
        else:
            inp_h, out_h = inp.cpu(), torch.empty((nrecv, _REC), dtype=torch.float64)
            try:
                dist.all_to_all_single(out_h, inp_h, recv_counts, in_splits, group=self.group)
            except RuntimeError:
                # gloo builds without alltoall: emulate with an all-gather of (padded) send buffers
                mx = int(mat.sum(dim=1).max())
                pad = torch.zeros((mx, _REC), dtype=torch.float64)
                pad[: inp_h.shape[0]] = inp_h
                bufs = [torch.zeros((mx, _REC), dtype=torch.float64) for _ in range(ws)]
                dist.all_gather(bufs, pad, group=self.group)
                parts = []
                for r in range(ws):
                    off = int(mat[r, : self.rank].sum())
                    parts.append(bufs[r][off: off + int(mat[r, self.rank])])
                out_h = torch.cat(parts) if parts else out_h
            out.copy_(out_h)
        return out, nrecv

    # ---- particle hand-over ----------------------------------------------------------------------------------
    def set_particles(self, pos, mass, ptype, old_acc=None, active=None, ids=None):
        super().set_particles(pos, mass, ptype, old_acc=old_acc, active=active)
        if ids is not None:
            ids = np.ascontiguousarray(ids, dtype=np.int64)
            self._check(self._L.ngravs_dd_set_ids(self._h, ids.ctypes.data, 0), "ngravs_dd_set_ids")

    def num_local(self):
        return int(self._L.ngravs_dd_num_local(self._h))

    def local_ids(self):
        out = np.zeros(self.num_local(), dtype=np.int64)
        self._check(self._L.ngravs_dd_get_ids(self._h, out.ctypes.data, 0), "ngravs_dd_get_ids")
        return out

    # ---- the step ------------------------------------------------------------------------------------------------
    def domain_Decomposition(self):
        dist, L = self.dist, self._L
        lo, hi = np.zeros(3), np.zeros(3)
        self._check(L.ngravs_dd_local_extent(self._h, lo.ctypes.data, hi.ctypes.data), "ngravs_dd_local_extent")
        lo = self._allreduce_host(lo, dist.ReduceOp.MIN)
        hi = self._allreduce_host(hi, dist.ReduceOp.MAX)
        self._check(L.ngravs_dd_set_extent(self._h, lo.ctypes.data, hi.ctypes.data), "ngravs_dd_set_extent")
        level = self._choose_level()
        ncell = 1 << (3 * level)
        hist = np.zeros(ncell, dtype=np.int64)
        self._check(L.ngravs_dd_histogram(self._h, level, hist.ctypes.data), "ngravs_dd_histogram")
        hist = self._allreduce_host(hist, dist.ReduceOp.SUM)
        owner_ph = cut_curve(hist, self.world)
        if self._ph is None or self._ph[0] != level:
            self._ph = (level, _ph_of_xyz(level))
        owner_xyz = np.ascontiguousarray(owner_ph[self._ph[1]].reshape(-1), dtype=np.int32)
        self.owner_ph, self.level_used = owner_ph, level
        counts = (C.c_int64 * 65)()
        ptr, nrec = C.c_void_p(), C.c_int64()
        for what in (0, 1):
            self._check(L.ngravs_dd_pack(self._h, what, level, owner_ph.ctypes.data, owner_xyz.ctypes.data, self.world, self.rank,
                                         counts, C.byref(ptr), C.byref(nrec)), "ngravs_dd_pack")
            recv, nrecv = self._alltoallv([counts[r] for r in range(self.world)], ptr.value, nrec.value)
            self._torch().cuda.synchronize()
            fn = L.ngravs_dd_apply_migration if what == 0 else L.ngravs_dd_set_halo
            self._check(fn(self._h, C.c_void_p(recv.data_ptr()), C.c_int64(nrecv)), "dd unpack")
            self.timings["migrated" if what == 0 else "halo"] = nrecv
        self.n = self.num_local()
        super().domain_Decomposition()

    def pmforce_periodic(self):
        torch, dist, L = self._torch(), self.dist, self._L
        self._check(L.ngravs_pm_deposit(self._h), "ngravs_pm_deposit")
        ptr, cnt = C.c_void_p(), C.c_int64()
        self._check(L.ngravs_pm_density(self._h, C.byref(ptr), C.byref(cnt)), "ngravs_pm_density")
        rho = torch.as_tensor(_DevArray(ptr.value, (cnt.value,)), device=self._dev())
        if self.backend == "nccl":
            dist.all_reduce(rho, group=self.group)
        else:
            h = rho.cpu()
            dist.all_reduce(h, group=self.group)
            rho.copy_(h)
        torch.cuda.synchronize()
        self._check(L.ngravs_pm_finish(self._h), "ngravs_pm_finish")

    def compute_accelerations(self, pm_step=True):
        self.domain_Decomposition()
        if pm_step and self.cfg.pmgrid:
            self.pmforce_periodic()
        self.gravity_tree()

    def get_accel(self, want_pm=False):
        """rows of this task's OWN particles (ids from local_ids()); halo rows are dropped"""
        nl = self.num_local()
        total = nl + int(self.timings.get("halo", 0))
        self.n = total
        res = super().get_accel(want_pm=want_pm)
        self.n = nl
        return tuple(r[:nl] if r is not None else None for r in res)
