"""gadget-2.0.7-ngravs_amd -- host-side mirror of the reference's gravity entry points over the
C ABI of libngravs_hip.so (include/ngravs_hip.h).

The directory name is not an importable identifier; load it with `__graft_entry__.load_package()`
(importlib, alias `ngravs_amd`).  The names below follow the reference (proto.h): an `Engine`
plays the role of the globals P[]/All/TypeToGrav[] for one MPI task and exposes
domain_Decomposition(), force_treebuild(), gravity_tree(), pmforce_periodic(),
compute_accelerations().  There is NO CPU fallback: if the HIP library or a GPU is missing every
entry point raises (the oracle under oracle/ is test infrastructure and is never imported here).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import abi, ic  # noqa: F401
from .abi import (Config, Particles, Stats, make_config, WALK_GROUP, WALK_STRICT,  # noqa: F401
                  LAW_NONE, LAW_NEWTON, LAW_NEG_NEWTON, LAW_YUKAWA, LAW_COLOYUK, LAW_BAMBAM, LAW_SOURCEBAM, LAW_TARGETBAM)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libngravs_hip.so")
_LIB = None

# every symbol include/ngravs_hip.h declares (tests/test_abi.py checks the .so exports all of them)
EXPORTS = [
    "ngravs_abi_version", "ngravs_build_info", "ngravs_config_default", "ngravs_create", "ngravs_destroy",
    "ngravs_set_fatal_handler", "ngravs_set_opening", "ngravs_set_walk_mode", "ngravs_set_softening", "ngravs_dd_record_bytes", "ngravs_get_config", "ngravs_set_tuning",
    "ngravs_memcpy", "ngravs_device_alloc", "ngravs_device_free",
    "ngravs_set_particles",
    "ngravs_set_old_acc", "ngravs_update_particles", "ngravs_force_update_tree", "ngravs_domain_decomposition", "ngravs_discard_grav_pm",
    "ngravs_force_treebuild", "ngravs_gravity_tree",
    "ngravs_pmforce_periodic", "ngravs_compute_accelerations", "ngravs_get_accel", "ngravs_get_stats", "ngravs_walk_unopened",
    "ngravs_get_domain", "ngravs_get_keys", "ngravs_get_order", "ngravs_get_shard", "ngravs_last_error",
    "ngravs_peano_hilbert_key", "ngravs_peano_keys", "ngravs_shortrange_table", "ngravs_direct_sum", "ngravs_direct_sum_targets",
    "ngravs_dd_num_local", "ngravs_dd_local_extent", "ngravs_dd_set_extent", "ngravs_get_domain_extent", "ngravs_dd_set_toptree",
    "ngravs_dd_get_toptree", "ngravs_dd_peano_order", "ngravs_dd_leaf_sums", "ngravs_dd_target_bounds", "ngravs_dd_keep_margin", "ngravs_dd_pack", "ngravs_dd_get_dest",
    "ngravs_dd_pack_leaves", "ngravs_dd_set_top", "ngravs_dd_recv_buffer", "ngravs_dd_apply_migration", "ngravs_dd_set_halo",
    "ngravs_dd_leaf_sums_kept", "ngravs_dd_pack_leaves_kept", "ngravs_dd_refresh_halo", "ngravs_dd_update_top", "ngravs_dd_get_kept",
    "ngravs_dd_set_ids", "ngravs_dd_get_ids",
    "ngravs_pm_slab_begin", "ngravs_pm_slab_pack", "ngravs_pm_slab_unpack", "ngravs_pm_slab_bytes",
]
# include/ngravs_host.h (plain-C multi-task drivers over a communicator vtable, linked into the same library)
HOST_EXPORTS = ["ngravs_host_comm_selftest", "ngravs_host_kept_step", "ngravs_host_toptree_borrow", "ngravs_host_domain_decomposition", "ngravs_host_domain_owners", "ngravs_host_domain_halo",
                "ngravs_host_plan_free", "ngravs_host_pmforce_periodic", "ngravs_host_compute_accelerations", "ngravs_host_split",
                "ngravs_host_pm_seconds", "ngravs_host_toptree_init", "ngravs_host_toptree_from_children", "ngravs_host_toptree_adapt",
                "ngravs_host_toptree_free", "ngravs_host_import_request", "ngravs_host_import_request_margin"]
# include/ngravs_comm_rccl.h (libngravs_rccl.so: the communicator vtable over RCCL, plain C)
RCCL_LIB_PATH = os.path.join(_HERE, "libngravs_rccl.so")
RCCL_EXPORTS = ["ngravs_rccl_selftest", "ngravs_rccl_unique_id", "ngravs_rccl_create", "ngravs_rccl_fill", "ngravs_rccl_destroy", "ngravs_rccl_stats",
                "ngravs_rccl_last_error", "ngravs_rccl_world", "ngravs_rccl_barrier", "ngravs_rccl_set_timeout"]


class NgravsError(RuntimeError):
    pass


def build(verbose=False):
    """hipcc --offload-arch=gfx950 build of csrc/ into libngravs_hip.so (in-tree)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise NgravsError("libngravs_hip.so is not built (run __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.ngravs_build_info.restype = C.c_char_p
        L.ngravs_last_error.restype = C.c_char_p
        L.ngravs_last_error.argtypes = [C.c_void_p]
        L.ngravs_peano_hilbert_key.restype = C.c_int64
        L.ngravs_peano_hilbert_key.argtypes = [C.c_int] * 4
        L.ngravs_force_treebuild.restype = C.c_int64
        L.ngravs_force_treebuild.argtypes = [C.c_void_p]
        for name in ("ngravs_destroy", "ngravs_domain_decomposition", "ngravs_gravity_tree",
                     "ngravs_pmforce_periodic", "ngravs_discard_grav_pm"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.ngravs_compute_accelerations.argtypes = [C.c_void_p, C.c_int]
        L.ngravs_set_opening.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.ngravs_set_walk_mode.argtypes = [C.c_void_p, C.c_int]
        L.ngravs_set_softening.argtypes = [C.c_void_p, C.c_void_p]
        L.ngravs_dd_record_bytes.restype = C.c_int64
        L.ngravs_dd_record_bytes.argtypes = [C.c_void_p, C.c_int]
        L.ngravs_set_particles.argtypes = [C.c_void_p, C.c_void_p]
        L.ngravs_set_old_acc.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        L.ngravs_get_accel.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                       C.c_void_p, C.c_int64, C.c_int, C.c_int]
        L.ngravs_set_tuning.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        L.ngravs_memcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        L.ngravs_get_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.ngravs_get_domain.argtypes = [C.c_void_p, C.c_void_p]
        L.ngravs_get_keys.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.ngravs_get_order.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.ngravs_get_shard.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ngravs_peano_keys.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_double, C.c_int, C.c_void_p]
        L.ngravs_shortrange_table.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ngravs_direct_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.ngravs_direct_sum_targets.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.ngravs_dd_local_extent.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ngravs_dd_set_extent.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ngravs_get_config.argtypes = [C.c_void_p, C.c_void_p]
        L.ngravs_host_split.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_void_p]
        L.ngravs_dd_pack.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ngravs_host_toptree_init.argtypes = [C.c_void_p, C.c_int]
        L.ngravs_host_toptree_from_children.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.ngravs_host_toptree_adapt.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p]
        L.ngravs_host_toptree_free.argtypes = [C.c_void_p]
        L.ngravs_host_toptree_free.restype = None
        L.ngravs_host_import_request.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ngravs_dd_apply_migration.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.ngravs_dd_set_halo.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.ngravs_dd_set_ids.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.ngravs_dd_get_ids.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _LIB = L
    return _LIB


def shard_range(n, rank, world_size):
    """Target shard of `rank`: positions [first, first+count) of the Peano order, cut on wave (64) boundaries so
    that target groups are identical for every world size (mirrors ngravs_domain_decomposition in capi.hip; the
    role of DomainMyStart/DomainMyLast after domain_findSplit, reference domain.c:347-456)."""
    lo = (n * rank) // world_size
    hi = (n * (rank + 1)) // world_size
    lo = (lo // 64) * 64
    hi = n if rank + 1 == world_size else (hi // 64) * 64
    return lo, hi - lo


def peano_hilbert_key(x, y, z, bits):
    """peano_hilbert_key() (reference peano.c:356-398), host implementation of the library."""
    return int(lib().ngravs_peano_hilbert_key(int(x), int(y), int(z), int(bits)))


def shortrange_table(cfg):
    """shortrange_fourier_force/pot[target][source][NTAB] (reference forcetree.c:3246-3403)."""
    ng = cfg.n_gravs
    force = np.zeros((ng, ng, abi.NTAB))
    pot = np.zeros((ng, ng, abi.NTAB))
    rc = lib().ngravs_shortrange_table(C.byref(cfg), force.ctypes.data, pot.ctypes.data)
    if rc != 0:
        raise NgravsError("ngravs_shortrange_table: %d" % rc)
    return force, pot


def _ptr(a):
    return a.ctypes.data if a is not None else None


class Engine:
    """One task's gravity state: the replacement for the reference's globals on this path."""

    def __init__(self, cfg):
        self.cfg = cfg
        self._h = C.c_void_p()
        rc = lib().ngravs_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise NgravsError("ngravs_create failed with status %d (no HIP device or bad wiring)" % rc)
        self._keep = []
        self.n = 0

    def close(self):
        if self._h:
            lib().ngravs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = lib().ngravs_last_error(self._h)
            raise NgravsError("%s failed: status %d (%s)" % (what, rc, msg.decode() if msg else ""))

    # -- P[] hand-over ------------------------------------------------------------------------
    def _host_columns(self, pos, mass, ptype, old_acc, active, grav_pm, grav_cost=None):
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        mass = np.ascontiguousarray(mass, dtype=np.float64)
        ptype = np.ascontiguousarray(ptype, dtype=np.int32)
        p = Particles()
        p.n = len(pos)
        p.pos, p.pos_stride = pos.ctypes.data, 24
        p.mass, p.mass_stride = mass.ctypes.data, 8
        p.type, p.type_stride = ptype.ctypes.data, 4
        keep = [pos, mass, ptype]
        if old_acc is not None:
            old_acc = np.ascontiguousarray(old_acc, dtype=np.float64)
            p.old_acc, p.old_acc_stride = old_acc.ctypes.data, 8
            keep.append(old_acc)
        if active is not None:
            active = np.ascontiguousarray(active, dtype=np.uint8)
            p.active, p.active_stride = active.ctypes.data, 1
            keep.append(active)
        if grav_pm is not None:
            grav_pm = np.ascontiguousarray(grav_pm, dtype=np.float64)
            p.grav_pm, p.grav_pm_stride = grav_pm.ctypes.data, 24
            keep.append(grav_pm)
        if grav_cost is not None:
            grav_cost = np.ascontiguousarray(grav_cost, dtype=np.float32)
            p.grav_cost, p.grav_cost_stride = grav_cost.ctypes.data, 4
            keep.append(grav_cost)
        p.on_device = 0
        self._keep = keep
        return p

    def set_particles(self, pos, mass, ptype, old_acc=None, active=None, grav_pm=None, grav_cost=None):
        """numpy (host) columns (the fields of P[] the path reads); grav_pm = P[].GravPM of the last PM step, if any;
        grav_cost = P[].GravCost (the work weight of the multi-task domain cut)"""
        p = self._host_columns(pos, mass, ptype, old_acc, active, grav_pm, grav_cost)
        self.n = p.n
        self._check(lib().ngravs_set_particles(self._h, C.byref(p)), "ngravs_set_particles")

    def update_particles(self, pos, mass, ptype, old_acc=None, active=None):
        """same particles, new positions / OldAcc / active flags: the decomposition and the tree topology are kept
        (drifted tree, TreeDomainUpdateFrequency > 0); gravity_tree() / pmforce_periodic() refit the nodes first"""
        p = self._host_columns(pos, mass, ptype, old_acc, active, None)
        self._check(lib().ngravs_update_particles(self._h, C.byref(p)), "ngravs_update_particles")

    def force_update_tree(self):
        self._check(lib().ngravs_force_update_tree(self._h), "ngravs_force_update_tree")

    def set_particles_device(self, n, pos_ptr, mass_ptr, type_ptr, old_acc_ptr=None, active_ptr=None, grav_pm_ptr=None):
        """HIP device pointers (e.g. torch tensors' data_ptr()): zero-copy hand-over."""
        p = Particles()
        p.n = n
        p.pos, p.pos_stride = pos_ptr, 24
        p.mass, p.mass_stride = mass_ptr, 8
        p.type, p.type_stride = type_ptr, 4
        if old_acc_ptr:
            p.old_acc, p.old_acc_stride = old_acc_ptr, 8
        if active_ptr:
            p.active, p.active_stride = active_ptr, 1
        if grav_pm_ptr:
            p.grav_pm, p.grav_pm_stride = grav_pm_ptr, 24
        p.on_device = 1
        self.n = n
        self._check(lib().ngravs_set_particles(self._h, C.byref(p)), "ngravs_set_particles")

    def set_old_acc(self, old_acc):
        old_acc = np.ascontiguousarray(old_acc, dtype=np.float64)
        self._check(lib().ngravs_set_old_acc(self._h, old_acc.ctypes.data, 8, 0), "ngravs_set_old_acc")

    def set_old_acc_device(self, ptr):
        """OldAcc from a HIP device pointer (N contiguous doubles, caller order)"""
        self._check(lib().ngravs_set_old_acc(self._h, ptr, 8, 1), "ngravs_set_old_acc")

    def get_old_acc_device(self, ptr):
        """write OldAcc (caller order) to a HIP device pointer"""
        self._check(lib().ngravs_get_accel(self._h, None, 0, None, 0, ptr, 8, None, 0, 1, 0), "ngravs_get_accel")

    def get_accel_device(self, acc_ptr=None, pm_ptr=None, old_ptr=None, cost_ptr=None, only_active=False):
        self._check(lib().ngravs_get_accel(self._h, acc_ptr, 24, pm_ptr, 24, old_ptr, 8, cost_ptr, 4, 1, int(only_active)),
                    "ngravs_get_accel")

    def walk_unopened(self):
        """top leaves the last group walk wanted opened but used as monopoles because they were not imported (multi-task trees)"""
        v = C.c_int64(0)
        self._check(lib().ngravs_walk_unopened(self._h, C.byref(v)), "ngravs_walk_unopened")
        return int(v.value)

    def set_opening(self, theta, err_tol_force_acc):
        self._check(lib().ngravs_set_opening(self._h, theta, err_tol_force_acc), "ngravs_set_opening")
        self.cfg.err_tol_theta = theta
        self.cfg.err_tol_force_acc = err_tol_force_acc

    def set_softening(self, force_softening):
        """All.ForceSoftening[6] after set_softenings() (gravtree.c:468-518): comoving runs change it every step"""
        fs = np.ascontiguousarray(force_softening, dtype=np.float64)
        assert fs.shape == (6,)
        self._check(lib().ngravs_set_softening(self._h, fs.ctypes.data), "ngravs_set_softening")
        for t in range(6):
            self.cfg.force_softening[t] = fs[t]

    def set_walk_mode(self, mode):
        self._check(lib().ngravs_set_walk_mode(self._h, mode), "ngravs_set_walk_mode")

    def set_tuning(self, **kw):
        """ngravs_set_tuning(): e.g. set_tuning(walk_lcap=1024, walk_compact=0)"""
        for k, v in kw.items():
            self._check(lib().ngravs_set_tuning(self._h, k.encode(), float(v)), "ngravs_set_tuning(%s)" % k)

    # -- the reference's entry points (proto.h names) ---------------------------------------------
    def domain_Decomposition(self):
        self._check(lib().ngravs_domain_decomposition(self._h), "domain_Decomposition")

    def force_treebuild(self):
        nn = lib().ngravs_force_treebuild(self._h)
        if nn < 0:
            self._check(int(nn), "force_treebuild")
        return int(nn)

    def gravity_tree(self):
        self._check(lib().ngravs_gravity_tree(self._h), "gravity_tree")

    def pmforce_periodic(self):
        self._check(lib().ngravs_pmforce_periodic(self._h), "pmforce_periodic")

    def compute_accelerations(self, pm_step=True):
        self._check(lib().ngravs_compute_accelerations(self._h, 1 if pm_step else 0), "compute_accelerations")

    # -- results -------------------------------------------------------------------------------------
    def get_accel(self, want_pm=False, into=None):
        """(GravAccel[N,3], OldAcc[N], GravCost[N]) [+ GravPM[N,3]] in the caller's particle order.
        into=(acc, old, cost): write ONLY the rows of active particles into these existing arrays, as the reference
        does with P[] (gravtree.c:318-341); otherwise fresh arrays, rows that were not walked read 0 / the input OldAcc."""
        n = self.n
        if into is not None:
            acc, old, cost = into
            assert acc.dtype == np.float64 and old.dtype == np.float64 and cost.dtype == np.float32
            assert acc.flags.c_contiguous and old.flags.c_contiguous and cost.flags.c_contiguous
        else:
            acc = np.zeros((n, 3))
            old = np.zeros(n)
            cost = np.zeros(n, dtype=np.float32)
        pm = np.zeros((n, 3)) if want_pm else None
        self._check(lib().ngravs_get_accel(self._h, acc.ctypes.data, 24, _ptr(pm), 24, old.ctypes.data, 8,
                                           cost.ctypes.data, 4, 0, 1 if into is not None else 0), "ngravs_get_accel")
        return (acc, old, cost, pm) if want_pm else (acc, old, cost)

    def get_pm(self):
        pm = np.zeros((self.n, 3))
        self._check(lib().ngravs_get_accel(self._h, None, 0, pm.ctypes.data, 24, None, 0, None, 0, 0, 0), "ngravs_get_accel")
        return pm

    def stats(self):
        s = Stats()
        self._check(lib().ngravs_get_stats(self._h, C.byref(s)), "ngravs_get_stats")
        return s

    def domain(self):
        d = np.zeros(8)
        self._check(lib().ngravs_get_domain(self._h, d.ctypes.data), "ngravs_get_domain")
        return d

    def keys(self):
        k = np.zeros(self.n, dtype=np.int64)
        self._check(lib().ngravs_get_keys(self._h, k.ctypes.data, 0), "ngravs_get_keys")
        return k

    def order(self):
        o = np.zeros(self.n, dtype=np.int32)
        self._check(lib().ngravs_get_order(self._h, o.ctypes.data, 0), "ngravs_get_order")
        return o

    def shard(self):
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(lib().ngravs_get_shard(self._h, C.byref(a), C.byref(b)), "ngravs_get_shard")
        return int(a.value), int(b.value)

    def peano_keys(self, pos, corner, fac, bits=18):
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        corner = np.ascontiguousarray(corner, dtype=np.float64)
        out = np.zeros(len(pos), dtype=np.int64)
        self._check(lib().ngravs_peano_keys(self._h, pos.ctypes.data, len(pos), corner.ctypes.data, float(fac), bits,
                                            out.ctypes.data), "ngravs_peano_keys")
        return out

    def direct_sum_targets(self, pos, ptype, mass=None):
        """partial direct sums of explicit targets over this task's OWN particles (distributed gravity_forcetest): add over tasks"""
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        ptype = np.ascontiguousarray(ptype, dtype=np.int32)
        mass = np.ascontiguousarray(mass, dtype=np.float64) if mass is not None else None
        out = np.zeros((len(pos), 3))
        self._check(lib().ngravs_direct_sum_targets(self._h, pos.ctypes.data, _ptr(mass), ptype.ctypes.data, len(pos), out.ctypes.data),
                    "ngravs_direct_sum_targets")
        return out

    def direct_sum(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        out = np.zeros((len(idx), 3))
        self._check(lib().ngravs_direct_sum(self._h, idx.ctypes.data, len(idx), out.ctypes.data), "ngravs_direct_sum")
        return out
