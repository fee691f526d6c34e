"""ctypes front-end of oracle/libngravs_oracle.so (TEST INFRASTRUCTURE ONLY: may be imported from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from the product package).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libngravs_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_peano_hilbert_key.restype = C.c_int64
        L.orc_peano_hilbert_key.argtypes = [C.c_int] * 4
        L.orc_tree_build.restype = C.c_void_p
        L.orc_tree_numnodes.restype = C.c_int64
        L.orc_tree_numnodes.argtypes = [C.c_void_p]
        L.orc_tree_ntopleaves.argtypes = [C.c_void_p]
        L.orc_tree_free.argtypes = [C.c_void_p]
        L.orc_law_eval.restype = C.c_double
        L.orc_law_eval.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def peano_key(x, y, z, bits):
    return int(lib().orc_peano_hilbert_key(int(x), int(y), int(z), int(bits)))


def domain_extent(pos):
    pos = _f64(pos)
    dom = np.zeros(8)
    lib().orc_domain_extent(_p(pos), C.c_int64(len(pos)), _p(dom))
    return dom


def keys(pos, dom):
    pos = _f64(pos)
    out = np.zeros(len(pos), dtype=np.int64)
    lib().orc_keys(_p(pos), C.c_int64(len(pos)), _p(_f64(dom)), _p(out))
    return out


def peano_order(cfg, key, ptype):
    key = np.ascontiguousarray(key, dtype=np.int64)
    ptype = _i32(ptype)
    out = np.zeros(len(key), dtype=np.int32)
    lib().orc_peano_order(C.byref(cfg), _p(key), _p(ptype), C.c_int64(len(key)), _p(out))
    return out


def direct_shortrange(cfg, pos, mass, ptype, idx, table, reach, nthreads=0):
    """test instrumentation: sum of the reference's short-range pair interaction over every particle within `reach`"""
    pos, mass, ptype, idx, table = _f64(pos), _f64(mass), _i32(ptype), _i32(idx), _f64(table)
    acc = np.zeros((len(idx), 3))
    nint = np.zeros(len(idx), dtype=np.int32)
    lib().orc_direct_shortrange(C.byref(cfg), _p(pos), _p(mass), _p(ptype), C.c_int64(len(pos)), _p(idx), C.c_int64(len(idx)), _p(table),
                                C.c_double(reach), _p(acc), _p(nint), C.c_int(nthreads))
    return acc, nint


def toptree_count(key):
    key = np.ascontiguousarray(key, dtype=np.int64)
    nl = C.c_int(0)
    nn = lib().orc_toptree_count(_p(key), C.c_int64(len(key)), C.byref(nl))
    return int(nn), int(nl.value)


class Tree:
    """force_treebuild() on the given particle order; keeps the arrays alive."""

    def __init__(self, cfg, pos, mass, ptype, dom=None):
        self.cfg = cfg
        self.pos, self.mass, self.type = _f64(pos), _f64(mass), _i32(ptype)
        self.dom = domain_extent(self.pos) if dom is None else _f64(dom)
        self.n = len(self.pos)
        self.h = lib().orc_tree_build(C.byref(cfg), _p(self.pos), _p(self.mass), _p(self.type),
                                      C.c_int64(self.n), _p(self.dom))
        self.h = C.c_void_p(self.h)

    def drift(self, newpos, vel, dt):
        """the reference's dynamic tree update (predict.c:79-91 node drift + force_update_len, forcetree.c:1005-1085)"""
        self.pos = _f64(newpos)
        self._vel = _f64(vel)
        lib().orc_tree_drift.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        lib().orc_tree_drift(self.h, C.byref(self.cfg), _p(self.pos), _p(self._vel), C.c_double(dt))

    def drift_kicked(self, newpos, vel, dv, dt):
        """the same after node kicks (timestep.c:331-344): node velocities from `vel` (the build), every particle kicked by dv"""
        self.pos = _f64(newpos)
        self._vel, self._dv = _f64(vel), _f64(dv)
        lib().orc_tree_drift_kicked.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        lib().orc_tree_drift_kicked(self.h, C.byref(self.cfg), _p(self.pos), _p(self._vel), _p(self._dv), C.c_double(dt))

    @property
    def numnodes(self):
        return int(lib().orc_tree_numnodes(self.h))

    @property
    def ntopleaves(self):
        return int(lib().orc_tree_ntopleaves(self.h))

    def node(self, i):
        out = np.zeros(4 + 4 * self.cfg.n_gravs)
        bf = C.c_int32(0)
        lib().orc_tree_get_node(self.h, C.c_int64(i), _p(out), C.byref(bf))
        return out, int(bf.value)

    def walk(self, old_acc=None, idx=None, table=None, cfg=None, nthreads=0):
        cfg = cfg if cfg is not None else self.cfg
        idx_a = _i32(idx) if idx is not None else None
        nt = len(idx_a) if idx_a is not None else self.n
        acc = np.zeros((nt, 3))
        nint = np.zeros(nt, dtype=np.int32)
        oa = _f64(old_acc) if old_acc is not None else None
        tb = _f64(table) if table is not None else None
        rc = lib().orc_walk(self.h, C.byref(cfg), _p(idx_a), C.c_int64(nt), _p(oa), _p(tb), _p(acc), _p(nint),
                            C.c_int(nthreads))
        if rc != 0:
            raise RuntimeError("orc_walk failed: %d" % rc)
        return acc, nint

    def walk_reach(self, idx, old_acc=None, table=None, cfg=None):
        """test instrumentation: per particle, the smallest node side through which it entered the walk of one of the targets
        idx (0: particle-particle interaction, inf: not reached except through larger nodes)"""
        cfg = cfg if cfg is not None else self.cfg
        idx_a = _i32(idx)
        reach = np.full(self.n, np.inf)
        oa = _f64(old_acc) if old_acc is not None else None
        tb = _f64(table) if table is not None else None
        rc = lib().orc_walk_reach(self.h, C.byref(cfg), _p(idx_a), C.c_int64(len(idx_a)), _p(oa), _p(tb), _p(reach))
        if rc != 0:
            raise RuntimeError("orc_walk_reach failed: %d" % rc)
        return reach

    def close(self):
        if self.h:
            lib().orc_tree_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def finish(cfg, acc, pm=None):
    """gravtree.c:318-341 -> (GravAccel*G, OldAcc)"""
    acc = _f64(acc).copy()
    old = np.zeros(len(acc))
    pm_a = _f64(pm) if pm is not None else None
    lib().orc_finish(C.byref(cfg), C.c_int64(len(acc)), _p(acc), _p(pm_a), _p(old))
    return acc, old


def shortrange_table(cfg):
    ng = cfg.n_gravs
    force = np.zeros((ng, ng, 2048))
    pot = np.zeros((ng, ng, 2048))
    lib().orc_shortrange_table(C.byref(cfg), _p(force), _p(pot))
    return force, pot


def pm_periodic(cfg, pos, mass, ptype):
    pos, mass, ptype = _f64(pos), _f64(mass), _i32(ptype)
    out = np.zeros((len(pos), 3))
    rc = lib().orc_pm_periodic(C.byref(cfg), _p(pos), _p(mass), _p(ptype), C.c_int64(len(pos)), _p(out))
    if rc != 0:
        raise RuntimeError("orc_pm_periodic failed: %d" % rc)
    return out


def direct(cfg, pos, mass, ptype, idx, nthreads=0):
    pos, mass, ptype, idx = _f64(pos), _f64(mass), _i32(ptype), _i32(idx)
    out = np.zeros((len(idx), 3))
    lib().orc_direct(C.byref(cfg), _p(pos), _p(mass), _p(ptype), C.c_int64(len(pos)), _p(idx),
                     C.c_int64(len(idx)), _p(out), C.c_int(nthreads))
    return out


def law_eval(cfg, which, law_id, a3, a4):
    return float(lib().orc_law_eval(C.byref(cfg), which, law_id, a3, a4))


EN1 = 65


def lattice_tables(cfg):
    """[tg][sg][3][65][65][65] force-correction tables (lattice_init), one Ewald/lattice sum per distinct law"""
    ng = cfg.n_gravs
    out = np.zeros((ng, ng, 3, EN1, EN1, EN1))
    cache = {}
    for a in range(ng):
        for b in range(ng):
            law = cfg.law_accel[a][b]
            if law not in cache:
                t = np.zeros((3, EN1, EN1, EN1))
                lib().orc_lattice_table(C.byref(cfg), C.c_int(law), _p(t))
                cache[law] = t
            out[a, b] = cache[law]
    return out


def lattice_walk(tree, acc, nint, lat, old_acc=None, idx=None, cfg=None, nthreads=0):
    """adds force_treeevaluate_lattice_correction to (acc, nint) in place"""
    cfg = cfg if cfg is not None else tree.cfg
    idx_a = _i32(idx) if idx is not None else None
    nt = len(idx_a) if idx_a is not None else tree.n
    oa = _f64(old_acc) if old_acc is not None else None
    lat = _f64(lat)
    lib().orc_lattice_walk(tree.h, C.byref(cfg), _p(idx_a), C.c_int64(nt), _p(oa), _p(lat), _p(acc), _p(nint), C.c_int(nthreads))
    return acc, nint


def direct_lattice(cfg, pos, mass, ptype, idx, lat, nthreads=0):
    pos, mass, ptype, idx, lat = _f64(pos), _f64(mass), _i32(ptype), _i32(idx), _f64(lat)
    out = np.zeros((len(idx), 3))
    lib().orc_direct_lattice(C.byref(cfg), _p(pos), _p(mass), _p(ptype), C.c_int64(len(pos)), _p(idx), C.c_int64(len(idx)),
                             _p(out), C.c_int(nthreads), _p(lat))
    return out
