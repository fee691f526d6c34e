"""The C-ABI library loads on a CPU-only box and exports every symbol include/ngravs_hip.h declares."""
import ctypes as C
import os
import re


def test_exports_every_declared_symbol(pkg, have_lib):
    hdr = open(os.path.join(os.path.dirname(pkg.__file__), "..", "include", "ngravs_hip.h")).read()
    declared = set(re.findall(r"\b(ngravs_[a-z0-9_]+)\s*\(", hdr)) - {"ngravs_fatal_fn"}
    assert declared, "no declarations found"
    assert declared == set(pkg.EXPORTS)
    for name in declared:
        assert hasattr(have_lib, name), name


def test_exports_the_c_host_layer(pkg, have_lib):
    hdr = open(os.path.join(os.path.dirname(pkg.__file__), "..", "include", "ngravs_host.h")).read()
    declared = set(re.findall(r"\b(ngravs_host_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(pkg.HOST_EXPORTS)
    for name in declared:
        assert hasattr(have_lib, name), name


def test_exports_the_rccl_communicator(pkg, have_lib):
    """libngravs_rccl.so (host/ngravs_comm_rccl.c) defines every function include/ngravs_comm_rccl.h declares (checked with nm: no
    RCCL call without a GPU), and it links against librccl, not against torch"""
    import subprocess
    hdr = open(os.path.join(os.path.dirname(pkg.__file__), "..", "include", "ngravs_comm_rccl.h")).read()
    declared = set(re.findall(r"\b(ngravs_rccl_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(pkg.RCCL_EXPORTS)
    assert os.path.exists(pkg.RCCL_LIB_PATH), "libngravs_rccl.so not built (__graft_entry__.build())"
    out = subprocess.run(["nm", "-D", "--defined-only", pkg.RCCL_LIB_PATH], capture_output=True, text=True, check=True).stdout
    defined = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert declared <= defined, declared - defined
    und = subprocess.run(["nm", "-D", "--undefined-only", pkg.RCCL_LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "ncclAllReduce" in und and "ncclSend" in und and "ncclRecv" in und and "ncclCommInitRank" in und


def test_host_split_balances_work_within_the_memory_bound(pkg, have_lib):
    """ngravs_host_split does the job of domain_findSplit + domain_shiftSplit (reference domain.c:347-544): contiguous runs of
    top leaves, every task gets leaves, the particle count of a task stays below max_load, and under that bound the largest work
    sum is as small as a contiguous cut allows (bottleneck-optimal: checked against the lower bounds)."""
    import numpy as np
    rng = np.random.default_rng(3)
    ncell, ntask = 4096, 8
    count = rng.integers(50, 150, ncell).astype(np.float64)
    work = count.copy()
    work[1000:1200] *= 40.0                         # a clump: few particles more, many interactions more
    owner_c = np.zeros(ncell, dtype=np.int32)
    owner_w = np.zeros(ncell, dtype=np.int32)
    maxload = 1.5 * count.sum() / ntask
    assert have_lib.ngravs_host_split(count.ctypes.data, None, ncell, ntask, maxload, owner_c.ctypes.data) == 0
    assert have_lib.ngravs_host_split(count.ctypes.data, work.ctypes.data, ncell, ntask, maxload, owner_w.ctypes.data) == 0
    for owner in (owner_c, owner_w):
        assert np.all(np.diff(owner) >= 0) and owner[0] == 0 and owner[-1] == ntask - 1      # contiguous runs, in task order
        assert len(np.unique(owner)) == ntask
        assert np.bincount(owner, weights=count).max() <= maxload
    wc = np.bincount(owner_c, weights=work)
    ww = np.bincount(owner_w, weights=work)
    cc = np.bincount(owner_c, weights=count)
    assert cc.max() / cc.mean() < 1.02                                                      # count cut: balanced by count
    assert ww.max() <= wc.max() and ww.max() / ww.mean() < 0.75 * wc.max() / wc.mean()       # work cut: better by work
    # the bottleneck cannot be lowered by moving one boundary leaf: the cut is optimal under the bound
    for t in range(ntask - 1):
        first_next = np.searchsorted(owner_w, t + 1)
        give, take = work[first_next - 1], work[first_next]
        assert max(ww[t] - give, ww[t + 1] + give) >= ww.max() - 1e-9 or np.bincount(owner_w, weights=count)[t + 1] + count[first_next - 1] > maxload \
            or ww[t] < ww.max() - 1e-9
        assert max(ww[t] + take, ww[t + 1] - take) >= min(ww.max(), max(ww[t], ww[t + 1])) - 1e-9 or \
            np.bincount(owner_w, weights=count)[t] + count[first_next] > maxload
    # an impossible memory bound is reported, not silently violated
    assert have_lib.ngravs_host_split(count.ctypes.data, None, ncell, ntask, 0.5 * count.sum() / ntask, owner_c.ctypes.data) == -1


def _tree_struct():
    class TopTree(C.Structure):
        _fields_ = [("nnode", C.c_int32), ("nleaf", C.c_int32), ("depth", C.c_int32), ("reserved", C.c_int32),
                    ("child", C.POINTER(C.c_int32)), ("level", C.POINTER(C.c_int32)), ("xyz", C.POINTER(C.c_int32)),
                    ("leaf", C.POINTER(C.c_int32)), ("node_of_leaf", C.POINTER(C.c_int32))]
    return TopTree


def leaf_counts(tree, key21):
    """particles per top leaf (curve order) from 63-bit Peano keys: the leaves are consecutive key ranges"""
    import numpy as np
    nn = tree.nnode
    child = np.ctypeslib.as_array(tree.child, (nn,))
    level = np.ctypeslib.as_array(tree.level, (nn,))
    leaf = np.ctypeslib.as_array(tree.leaf, (nn,))
    # start key of every node: descend
    start = np.zeros(nn, dtype=np.int64)
    for i in range(nn):
        if child[i] >= 0:
            for k in range(8):
                start[child[i] + k] = int(start[i]) + (k << (3 * (21 - int(level[i]) - 1)))
    nodes = np.flatnonzero(leaf >= 0)
    nodes = nodes[np.argsort(leaf[nodes])]
    lo = start[nodes]
    hi = lo.astype(np.uint64) + (np.uint64(1) << (3 * (21 - level[nodes])).astype(np.uint64))    # 2^63 for the root's end
    ks = np.sort(key21).astype(np.uint64)
    return (np.searchsorted(ks, hi) - np.searchsorted(ks, lo.astype(np.uint64))).astype(np.float64)


def build_toptree(lib, key21, thresh, start_level=1, max_level=18):
    """the top tree of a particle set by rounds of ngravs_host_toptree_adapt, as ngravs_host_domain_owners does with all-reduced
    counts"""
    TopTree = _tree_struct()
    t = TopTree()
    assert lib.ngravs_host_toptree_init(C.byref(t), start_level) == 0
    rounds = 0
    while True:
        cnt = leaf_counts(t, key21)
        assert cnt.sum() == len(key21)
        nxt = TopTree()
        unknown = lib.ngravs_host_toptree_adapt(C.byref(t), cnt.ctypes.data, float(thresh), max_level, C.byref(nxt))
        assert unknown >= 0
        rounds += 1
        if unknown == 0 and nxt.nnode in (0, t.nnode):
            lib.ngravs_host_toptree_free(C.byref(nxt))
            return t, cnt, rounds
        lib.ngravs_host_toptree_free(C.byref(t))
        t = nxt
        assert rounds < 40


def test_top_tree_is_the_references_on_its_own_ic(pkg, have_lib, O, kats):
    """domain_determineTopTree on GalaxyCollision.IC with one task: 176 top leaves (recorded from the reference, SURVEY.md 8(c);
    the oracle's restatement reproduces it).  The product's rule -- rounds of ngravs_host_toptree_adapt on leaf counts, from any
    starting tree -- must arrive at the same tree: same number of nodes and leaves, every leaf within the threshold, every split
    node above it."""
    import numpy as np
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(pkg.__file__), "..", "tests"))
    from conftest import galaxy_ic
    ic = galaxy_ic(pkg)
    pos = ic["pos"]
    dom = O.domain_extent(pos)
    key18 = O.keys(pos, dom)
    ntop, nleaves = O.toptree_count(key18)
    assert nleaves == kats["galaxy_collision"]["ntopleaves"] == 176
    thresh = len(pos) / 20.0                                   # TotNumPart / (TOPNODEFACTOR * NTask), NTask = 1
    key21 = key18.astype(np.int64) << 9                        # the engine's 21-bit keys: the reference's 18-bit keys are their prefix
    for start in (1, 3, 5):                                    # from a coarser and from a finer tree than the answer
        t, cnt, rounds = build_toptree(have_lib, key21, thresh, start_level=start)
        child = np.ctypeslib.as_array(t.child, (t.nnode,))
        assert t.nleaf == nleaves and t.nnode == ntop, (start, t.nleaf, t.nnode, ntop, nleaves)
        assert cnt.max() <= thresh and (child >= 0).sum() * 8 + 1 == t.nnode
        print("start level %d: %d rounds -> %d nodes, %d leaves, largest leaf %d particles" % (start, rounds, t.nnode, t.nleaf, cnt.max()))
        have_lib.ngravs_host_toptree_free(C.byref(t))


def test_struct_sizes_match_header(pkg, have_lib):
    info = have_lib.ngravs_build_info().decode()
    sizes = dict(re.findall(r"sizeof\((\w+)\)=(\d+)", info))
    assert int(sizes["config"]) == C.sizeof(pkg.Config)
    assert int(sizes["particles"]) == C.sizeof(pkg.Particles)
    assert int(sizes["stats"]) == C.sizeof(pkg.Stats)
    assert have_lib.ngravs_abi_version() == pkg.abi.ABI_VERSION


def test_bad_wiring_is_rejected_without_a_gpu(pkg, have_lib):
    cfg = pkg.make_config(n_gravs=2, wiring="newton")
    cfg.law_accel[0][1] = pkg.LAW_YUKAWA          # asymmetric: violates Newton's third law probe
    h = C.c_void_p()
    assert have_lib.ngravs_create(C.byref(cfg), C.byref(h)) == -6
    cfg = pkg.make_config(n_gravs=1, pmgrid=32, periodic=0, box_size=1.0)
    assert have_lib.ngravs_create(C.byref(cfg), C.byref(h)) == -1   # non-periodic PM is disabled by ngravs
    cfg = pkg.make_config(n_gravs=1)
    cfg.n_gravs = 4
    assert have_lib.ngravs_create(C.byref(cfg), C.byref(h)) == -1


def test_top_tree_rule_on_degenerate_and_clustered_sets(pkg, have_lib):
    """ngravs_host_toptree_adapt: the rule "split while count > threshold" reaches a fixed point from any start, merges what a
    finer start split too far, stops at the key resolution for coincident particles, and every leaf of the result either obeys
    the threshold or sits at the deepest level."""
    import numpy as np
    rng = np.random.default_rng(8)
    n = 50000
    # a clump of coincident keys + a Gaussian blob + a uniform background (keys: 63-bit, 21 bits per dimension)
    blob = np.clip(rng.normal(0.3, 0.01, (n // 2, 3)), 0, 0.999999)
    back = rng.uniform(0, 1, (n // 2 - 500, 3))
    same = np.full((500, 3), 0.77)
    pos = np.concatenate([blob, back, same])
    ix = (pos * (1 << 21)).astype(np.int64)
    key21 = np.array([have_lib.ngravs_peano_hilbert_key(int(a), int(b), int(c), 21) for a, b, c in ix[::1]], dtype=np.int64)
    results = []
    for start in (1, 4, 6):
        t, cnt, rounds = build_toptree(have_lib, key21, 64.0, start_level=start)
        nn = t.nnode
        child = np.ctypeslib.as_array(t.child, (nn,)).copy()
        level = np.ctypeslib.as_array(t.level, (nn,)).copy()
        leaf = np.ctypeslib.as_array(t.leaf, (nn,)).copy()
        lv = level[np.argsort(np.where(leaf >= 0, leaf, 1 << 30))[: t.nleaf]]
        assert cnt.sum() == n
        heavy = cnt > 64.0
        assert np.all(lv[heavy] == 18) and heavy.sum() >= 1          # only the coincident clump, at the key resolution
        assert cnt[heavy].sum() >= 500
        # every split node holds more than the threshold (nothing was split too far and left that way)
        ncount = np.zeros(nn)
        for i in range(nn - 1, -1, -1):
            ncount[i] = cnt[leaf[i]] if child[i] < 0 else ncount[child[i]: child[i] + 8].sum()
        assert np.all(ncount[(child >= 0) & (np.arange(nn) > 0)] > 64.0)
        results.append((t.nnode, t.nleaf, child.tobytes()))
        print("start level %d: %d rounds -> %d nodes / %d leaves, depth %d" % (start, rounds, t.nnode, t.nleaf, t.depth))
        have_lib.ngravs_host_toptree_free(C.byref(t))
    assert results[0] == results[1] == results[2]                    # the rule has ONE fixed point
    # fewer leaves than tasks: the cut reports it instead of inventing owners
    owner = np.zeros(4, dtype=np.int32)
    c4 = np.array([5.0, 1.0, 1.0, 1.0])
    assert have_lib.ngravs_host_split(c4.ctypes.data, None, 4, 8, 0.0, owner.ctypes.data) == -1
    assert have_lib.ngravs_host_split(c4.ctypes.data, None, 4, 4, 0.0, owner.ctypes.data) == 0 and list(owner) == [0, 1, 2, 3]
