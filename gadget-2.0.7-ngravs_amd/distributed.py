"""Multi-task (one process per GPU) host layer.

The choreography itself -- domain_Decomposition() with the adaptive top tree, work-weighted cuts, particle migration and the
import of the top leaves a task may open (reference domain.c:62-330, 347-544, 695-795, 933-1138; forcetree.c:766-996;
gravtree.c:112-285) and pmforce_periodic() on the x-slab decomposed mesh (pm_periodic.c:204-790) -- is plain C inside
libngravs_hip.so (host/ngravs_host.c, include/ngravs_host.h), written over a communicator vtable.  This module only provides
that vtable:

  RcclComm   production: libngravs_rccl.so (host/ngravs_comm_rccl.c, include/ngravs_comm_rccl.h) -- ncclAllReduce /
             ncclAllGather / grouped ncclSend+ncclRecv straight on the library's device buffers, in C.  No Python runs inside a
             step.  torch.distributed is used ONCE, to broadcast the 128-byte ncclUniqueId (the reference glue does that with
             MPI_Bcast), so a Python host and a C host share one RCCL path.
  TorchComm  rehearsal: torch.distributed "gloo" through host copies (several tasks on one GPU, or no RCCL).

Collectives per step in the steady state (DistributedEngine.info.collectives, .pm_bytes):
  all-reduce   MIN of extent + target bounds + status + largest own count (10 f64); SUM of the per-leaf sums (NGRAVS_TOP_CW doubles per top leaf)
  all-gather   migration counts + import requests (one call); PM bricks' bounding boxes (7 i32)
  all-to-all-v imported leaves (56-byte records), the four mesh exchanges of the slab PM; migrating particles only when any move
"""
import ctypes as C

import numpy as np
import torch  # noqa: F401  (before libngravs_hip.so is loaded whenever possible: one HIP runtime per process, torch's own)

from . import Engine, NgravsError, lib

_I64P = C.POINTER(C.c_int64)
_ALLREDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int)
_ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
_ALLTOALLV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, _I64P, _I64P, C.c_void_p, _I64P, _I64P)


class Comm(C.Structure):
    """struct ngravs_comm (include/ngravs_host.h)"""
    _fields_ = [("rank", C.c_int32), ("size", C.c_int32), ("device_buffers", C.c_int32), ("reserved", C.c_int32),
                ("user", C.c_void_p), ("allreduce", _ALLREDUCE), ("allgather", _ALLGATHER), ("alltoallv", _ALLTOALLV),
                ("allreduce_dev", _ALLREDUCE)]


class DDInfo(C.Structure):
    """struct ngravs_dd_info (include/ngravs_host.h)"""
    _fields_ = [("n_topnodes", C.c_int32), ("n_topleaves", C.c_int32), ("n_local", C.c_int64), ("n_halo", C.c_int64),
                ("n_migrated_in", C.c_int64), ("work_balance", C.c_double), ("memory_balance", C.c_double),
                ("bytes_migration", C.c_double), ("bytes_halo", C.c_double), ("seconds", C.c_double * 8),
                ("toptree_rounds", C.c_int32), ("collectives", C.c_int32)]


class RcclComm:
    """struct ngravs_comm filled by libngravs_rccl.so (C): RCCL on the library's device buffers, no Python in the data path"""

    def __init__(self, device, group=None):
        import torch
        import torch.distributed as dist
        from . import RCCL_LIB_PATH
        self.rank, self.size = dist.get_rank(group), dist.get_world_size(group)
        self.backend = "rccl (libngravs_rccl.so)"
        self.error = None
        L = self._L = C.CDLL(RCCL_LIB_PATH)
        L.ngravs_rccl_last_error.restype = C.c_char_p
        L.ngravs_rccl_last_error.argtypes = [C.c_void_p]
        L.ngravs_rccl_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ngravs_rccl_fill.argtypes = [C.c_void_p, C.c_void_p]
        L.ngravs_rccl_destroy.argtypes = [C.c_void_p]
        L.ngravs_rccl_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.ngravs_rccl_world.argtypes = [C.c_void_p]
        ident = C.create_string_buffer(128)
        if self.rank == 0 and L.ngravs_rccl_unique_id(ident) != 0:
            raise NgravsError("ncclGetUniqueId failed")
        box = [ident.raw]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self._h = C.c_void_p()
        dev = device.index if hasattr(device, "index") else int(device)
        if L.ngravs_rccl_create(box[0], self.rank, self.size, dev, C.byref(self._h)) != 0:
            raise NgravsError("ngravs_rccl_create failed (one task per GPU: RCCL refuses two ranks of a communicator on one device)")
        self.c = Comm()
        L.ngravs_rccl_fill(self._h, C.byref(self.c))
        self.world_reported = int(L.ngravs_rccl_world(self._h))
        L.ngravs_rccl_selftest.argtypes = [C.c_void_p]

    def selftest(self):
        """every collective of the vtable once with known answers (ngravs_rccl_selftest); collective.  Returns "" or what differed"""
        return "" if self._L.ngravs_rccl_selftest(self._h) == 0 else (self.last_error() or "self test failed")

    def stats(self):
        calls, sec, byt = C.c_int64(0), C.c_double(0), C.c_double(0)
        self._L.ngravs_rccl_stats(self._h, C.byref(calls), C.byref(sec), C.byref(byt), 0)
        return int(calls.value), float(sec.value), float(byt.value)

    @property
    def seconds(self):
        return self.stats()[1]

    @property
    def calls(self):
        return self.stats()[0]

    def last_error(self):
        m = self._L.ngravs_rccl_last_error(self._h)
        return m.decode() if m else ""

    def close(self):
        if self._h:
            self._L.ngravs_rccl_destroy(self._h)
            self._h = C.c_void_p()


class _DevArray:
    """zero-copy view of library device memory for torch (CUDA array interface)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def _host_view(ptr, nbytes):
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(int(nbytes),)) if nbytes > 0 else np.zeros(0, np.uint8)


class TorchComm:
    """struct ngravs_comm backed by a torch.distributed process group"""

    def __init__(self, device, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group, self.device = torch, dist, group, device
        self.rank, self.size = dist.get_rank(group), dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.error = None
        self.seconds, self.calls = 0.0, 0
        self._cb = (_ALLREDUCE(self._allreduce), _ALLGATHER(self._allgather), _ALLTOALLV(self._alltoallv))
        self.c = Comm(self.rank, self.size, 1, 0, None, *self._cb, C.cast(None, _ALLREDUCE))   # no device-side reduction
        self.world_reported = self.size

    def close(self):
        pass

    def last_error(self):
        return repr(self.error) if self.error is not None else ""

    def _guard(self, fn, *a):
        import time
        t0 = time.perf_counter()
        try:
            fn(*a)
            return 0
        except Exception as e:          # never let an exception cross the C frame
            self.error = e
            return 1
        finally:
            self.seconds += time.perf_counter() - t0     # wall time inside collectives (incl. waiting for the slowest task)
            self.calls += 1

    def _allreduce(self, user, buf, count, dtype, op):
        def run():
            torch, dist = self.torch, self.dist
            h = _host_view(buf, 8 * count).view(np.float64 if dtype == 0 else np.int64)
            t = torch.from_numpy(h.copy())
            if self.backend == "nccl":
                t = t.to(self.device)
            dist.all_reduce(t, op={0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MIN, 2: dist.ReduceOp.MAX}[op], group=self.group)
            h[:] = t.cpu().numpy()
        return self._guard(run)

    def _allgather(self, user, send, recv, nbytes):
        def run():
            torch, dist = self.torch, self.dist
            t = torch.from_numpy(_host_view(send, nbytes).copy())
            if self.backend == "nccl":
                t = t.to(self.device)
            outs = [torch.empty_like(t) for _ in range(self.size)]
            dist.all_gather(outs, t, group=self.group)
            _host_view(recv, self.size * nbytes)[:] = torch.cat(outs).cpu().numpy()
        return self._guard(run)

    def _alltoallv(self, user, send, sbytes, sdispl, recv, rbytes, rdispl):
        def run():
            torch, dist, W = self.torch, self.dist, self.size
            sb, rb = [int(sbytes[r]) for r in range(W)], [int(rbytes[r]) for r in range(W)]
            ns, nr = sum(sb), sum(rb)      # blocks are contiguous in task order (displacements are prefix sums)
            inp = torch.as_tensor(_DevArray(send, ns), device=self.device) if ns else torch.empty(0, dtype=torch.uint8, device=self.device)
            out = torch.as_tensor(_DevArray(recv, nr), device=self.device) if nr else torch.empty(0, dtype=torch.uint8, device=self.device)
            if self.backend == "nccl":
                dist.all_to_all_single(out, inp, rb, sb, group=self.group)
            else:
                inp_h, out_h = inp.cpu(), torch.empty(nr, dtype=torch.uint8)
                try:
                    dist.all_to_all_single(out_h, inp_h, rb, sb, group=self.group)
                except RuntimeError:
                    # gloo builds without alltoall: emulate with all-gathers of the count matrix and the padded send buffers
                    cnt = torch.tensor(sb, dtype=torch.int64)
                    mats = [torch.zeros(W, dtype=torch.int64) for _ in range(W)]
                    dist.all_gather(mats, cnt, group=self.group)
                    mat = torch.stack(mats)
                    mx = int(mat.sum(dim=1).max())
                    pad = torch.zeros(max(mx, 1), dtype=torch.uint8)
                    pad[:ns] = inp_h
                    bufs = [torch.zeros(max(mx, 1), dtype=torch.uint8) for _ in range(W)]
                    dist.all_gather(bufs, pad, group=self.group)
                    parts = []
                    for r in range(W):
                        off = int(mat[r, : self.rank].sum())
                        parts.append(bufs[r][off: off + int(mat[r, self.rank])])
                    out_h = torch.cat(parts)
                if nr:
                    out.copy_(out_h)
            torch.cuda.synchronize(self.device)      # the library's unpack kernels run on its own stream
        return self._guard(run)


class DistributedEngine(Engine):
    def __init__(self, cfg, leaf_max=None, group=None, comm=None):
        """leaf_max: a top-tree node is split while it holds more particles (None: the reference's TotNumPart / (20 NTask), at
        most NGRAVS_TOPLEAF_MAX).  comm: "rccl" (C, libngravs_rccl.so) | "torch" | None = rccl when the process group's backend
        is nccl, else torch."""
        import torch
        import torch.distributed as dist
        cfg.rank, cfg.world_size = 0, 1          # the library sees its working set (own + imported) as a single task
        super().__init__(cfg)
        dev = torch.device("cuda", cfg.device)
        if comm is None:
            comm = "rccl" if dist.get_backend(group) == "nccl" else "torch"
        self.comm = RcclComm(dev, group) if comm == "rccl" else TorchComm(dev, group)
        self.group, self._dev = group, dev
        self.comm_note = ""
        if comm == "rccl":
            # the C communicator proves itself once; if it does not on ANY task, all tasks fall back together to the process group
            # (the same collectives through torch.distributed) and say so -- never a silent switch, never a hang inside a step
            why = self.comm.selftest()
            bad = torch.tensor([1 if why else 0], dtype=torch.int32, device=dev if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
            if int(bad.item()):
                import sys
                print("ngravs: libngravs_rccl.so self test failed on task %d (%s): falling back to torch.distributed collectives"
                      % (self.comm.rank, why or "on another task"), file=sys.stderr)
                self.comm.close()
                self.comm = TorchComm(dev, group)
                self.comm_note = "fallback: libngravs_rccl self test failed (%s)" % (why or "on another task")
        self.rank, self.world, self.backend = self.comm.rank, self.comm.size, self.comm.backend
        if self.world > 64:
            raise NgravsError("at most 64 tasks")
        self.leaf_max = leaf_max
        self.info = DDInfo()
        self.timings = {}
        self.reset_wall()
        L = self._L = lib()
        L.ngravs_dd_num_local.restype = C.c_int64
        L.ngravs_dd_num_local.argtypes = [C.c_void_p]
        L.ngravs_host_domain_decomposition.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
        L.ngravs_host_pmforce_periodic.argtypes = [C.c_void_p, C.c_void_p]
        L.ngravs_host_kept_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ngravs_pm_slab_bytes.argtypes = [C.c_void_p, C.c_void_p]

    def _host(self, rc, what):
        if self.comm.error is not None:
            e, self.comm.error = self.comm.error, None
            raise e
        if rc != 0 and self.comm.last_error():
            raise NgravsError("%s failed: status %d; communicator: %s" % (what, rc, self.comm.last_error()))
        self._check(rc, what)

    def close(self):
        super().close()
        if getattr(self, "comm", None) is not None:
            self.comm.close()
            self.comm = None

    # ---- particle hand-over ----------------------------------------------------------------------------------
    def set_particles(self, pos, mass, ptype, old_acc=None, active=None, ids=None, grav_pm=None, grav_cost=None):
        super().set_particles(pos, mass, ptype, old_acc=old_acc, active=active, grav_pm=grav_pm, grav_cost=grav_cost)
        if ids is not None:
            ids = np.ascontiguousarray(ids, dtype=np.int64)
            self._check(self._L.ngravs_dd_set_ids(self._h, ids.ctypes.data, 0), "ngravs_dd_set_ids")

    def num_local(self):
        return int(self._L.ngravs_dd_num_local(self._h))

    def local_ids(self):
        out = np.zeros(self.num_local(), dtype=np.int64)
        self._check(self._L.ngravs_dd_get_ids(self._h, out.ctypes.data, 0), "ngravs_dd_get_ids")
        return out

    # ---- the step (the reference's names) ----------------------------------------------------------------------------
    def domain_Decomposition(self):
        rc = self._L.ngravs_host_domain_decomposition(self._h, C.byref(self.comm.c), float(self.leaf_max or 0.0), 0.0, C.byref(self.info))
        self._host(rc, "ngravs_host_domain_decomposition")
        self.timings["migrated"], self.timings["halo"] = int(self.info.n_migrated_in), int(self.info.n_halo)
        self.n = self.num_local()

    def kept_step(self, pos, mass, ptype, old_acc=None, active=None):
        """A step that KEEPS the decomposition (domain.c:76 with All.TreeDomainUpdateFrequency > 0): the own rows (in the order of
        local_ids()) with their drifted positions; the imported copies are refreshed by their owners, the tree is refit, the top
        nodes get their global moments and sides again (ngravs_host_kept_step: two collectives).  Then gravity_tree() as usual."""
        self.n = self.num_local()
        self.update_particles(pos, mass, ptype, old_acc=old_acc, active=active)
        self._host(self._L.ngravs_host_kept_step(self._h, C.byref(self.comm.c), C.byref(self.info)), "ngravs_host_kept_step")

    def kept_walk_missed(self):
        """After gravity_tree() on a kept step: did the walk of ANY task want a top leaf that was never imported (ngravs_walk_unopened,
        all-reduced)?  The production walk used such a leaf as a monopole and counted it; the reference walk refuses instead.  True means:
        decompose (domain_Decomposition) and walk again -- what gadget_glue.c does."""
        import torch
        import torch.distributed as dist
        t = torch.tensor([int(self.walk_unopened())], dtype=torch.int64,
                         device=(self._dev if dist.get_backend(self.group) == "nccl" else "cpu"))
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item()) > 0

    def pmforce_periodic(self):
        self._host(self._L.ngravs_host_pmforce_periodic(self._h, C.byref(self.comm.c)), "ngravs_host_pmforce_periodic")

    def pm_seconds(self):
        """host wall-clock seconds of the stages of the last slab PM step (ngravs_host_pm_seconds)"""
        b = (C.c_double * 13)()
        self._L.ngravs_host_pm_seconds(b)
        return list(b)

    def pm_bytes(self):
        """payload this task sent to other tasks in the four mesh exchanges of the last PM step (bytes)"""
        b = (C.c_double * 4)()
        self._check(self._L.ngravs_pm_slab_bytes(self._h, b), "ngravs_pm_slab_bytes")
        return list(b)

    def compute_accelerations(self, pm_step=True):
        """One force computation; self.wall = host wall-clock seconds of its three stages and of the collectives inside them
        (decomposition incl. migration + tree-node import, PM incl. the four plane exchanges, tree walk), summed over the steps
        since the last reset_wall()."""
        import time
        w = self.wall
        c0, n0 = self.comm.seconds, self.comm.calls
        t0 = time.perf_counter()
        if pm_step and self.cfg.pmgrid:
            self._check(self._L.ngravs_discard_grav_pm(self._h), "ngravs_discard_grav_pm")    # recomputed below
        self.domain_Decomposition()
        t1 = time.perf_counter()
        c1 = self.comm.seconds
        if pm_step and self.cfg.pmgrid:
            self.pmforce_periodic()
        t2 = time.perf_counter()
        c2 = self.comm.seconds
        self.gravity_tree()
        t3 = time.perf_counter()
        w["steps"] += 1
        w["decomposition_s"] += t1 - t0
        w["decomposition_collectives_s"] += c1 - c0
        w["pm_s"] += t2 - t1
        w["pm_collectives_s"] += c2 - c1
        w["gravity_tree_s"] += t3 - t2
        w["collective_calls"] += self.comm.calls - n0

    def reset_wall(self):
        self.wall = {"steps": 0, "decomposition_s": 0.0, "decomposition_collectives_s": 0.0, "pm_s": 0.0, "pm_collectives_s": 0.0,
                     "gravity_tree_s": 0.0, "collective_calls": 0}

    def get_accel(self, want_pm=False, into=None):
        """rows of this task's OWN particles (ids from local_ids()); the library never delivers the imported copies"""
        self.n = self.num_local()
        return super().get_accel(want_pm=want_pm, into=into)

    def order(self):
        """Peano order of the working set: own rows [0, num_local()) and the imported copies behind them"""
        nl = self.num_local()
        self.n = nl + int(self.info.n_halo)
        try:
            return super().order()
        finally:
            self.n = nl
