#!/usr/bin/env python3
"""bench.py -- particle-steps/s of the gravity path (tree + PM) on MI355X.

Workload at N=1 (BASELINE.json configs[3], the configuration the metric is quoted on): 2^26 = 64 M
particles in a periodic box, N_GRAVS=2 (diagonal Newton, off-diagonal Newton+Yukawa "coloyuk",
YUKAWA_IMASS=60; SURVEY.md 8(d) wiring for C4), TreePM with PMGRID=512, ASMTH=1.25, RCUT=4.5,
NTAB=2048, softening eps = L/(40 N^(1/3)), ErrTolForceAcc=0.005 with the relative opening criterion
(the steady-state second pass of accel.c:48-52; OldAcc comes from an untimed theta=0.5 pass).

A "step" is one compute_accelerations(): domain extent + Peano keys + sort (domain_Decomposition),
pmforce_periodic, force_treebuild, gravity_tree (walk + OldAcc/G post-processing) for ALL particles,
inputs already resident in HBM.  value = particles * steps / wall time (max over ranks).

  --gpus N : one process per GPU, N tasks over RCCL (torch.distributed backend "nccl").  Started by the driver's
             `python -m torch.distributed.run ... bench.py --gpus N`, or by itself: without WORLD_SIZE in the environment
             `python bench.py --gpus N` starts its own N children through torch.distributed.run (the parent never touches
             a GPU) and relays rank 0's JSON line and exit code.
             Default decomposition (TreePM configs): the reference's own -- work-weighted Peano-Hilbert domains cut at
             top-tree leaves, particle migration, all-reduced top-leaf moments + import of the top-tree cells a task may
             open, x-slab decomposed PM with four plane exchanges; total work is fixed: "scaling": "strong".
             --decomp replicated (tree-only configs): every rank holds all particles, only the walk is sharded.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK = 8.0e12     # B/s, MI355X_MICROARCH.md


def walk_alg_bytes(n_gravs):
    # target 44 + result 28 + every source record once 36 + 0.5*(48+32g)
    return 108 + 0.5 * (48 + 32 * n_gravs)


def step_alg_bytes(n_gravs, cells_per_particle, pm=True):
    g = n_gravs
    b = 96 + (36 + 0.5 * (48 + 32 * g)) + walk_alg_bytes(g)
    if pm:
        b += 88 + 64 * g * cells_per_particle
    return b


# ngravs_dd_info.seconds[] (include/ngravs_host.h)
DD_STAGES = ["extent+leaf_sums+top_tree+cut", "migration", "of_the_first:leaf_sum_passes+allreduce", "import_decision_host",
             "count/request_allgather+pack", "import_exchange+unpack", "global_top", "local_decomposition"]

TRAFFIC_PROFILE = os.path.join("profiles", "r04_walk_traffic.json")


WALK_SOURCES = ("kernels_walk.hip", "kernels_eval.hip", "eval_asm.inc", "walk_device.hpp")


def walk_source_hash():
    """sha256 (first 16 hex digits) of the walk kernels' sources: ties a PMC profile to the code it was taken from"""
    import hashlib
    h = hashlib.sha256()
    for name in WALK_SOURCES:
        with open(os.path.join(ROOT, "gadget-2.0.7-ngravs_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def walk_traffic(args, n, world, launches):
    """HBM-side bytes per launch of the walk's evaluation kernel.  PMC counters cannot be collected inside this process:
    the figure comes from the committed rocprofv3 --pmc passes of THIS workload (tools/profile_c4.sh), and only while the
    kernel source is the one that was profiled -- otherwise traffic is null.  Returns (bytes or None, provenance)."""
    path = os.path.join(ROOT, TRAFFIC_PROFILE)
    prov = {"source": TRAFFIC_PROFILE, "walk_source_sha16": walk_source_hash()}
    if not (args.config == "c4" and n == (1 << 26) and world == 1 and args.walk == "group" and not args.tune):
        prov["status"] = "not the profiled workload"
        return None, prov
    if not os.path.exists(path):
        prov["status"] = "no profile committed"
        return None, prov
    with open(path) as f:
        d = json.load(f)
    prov["profiled_kernel"] = d.get("kernel")
    prov["profiled_walk_source_sha16"] = d.get("walk_source_sha16")
    if d.get("walk_source_sha16") != prov["walk_source_sha16"]:
        prov["status"] = "stale: the walk kernels' sources changed since the profile was taken"
        return None, prov
    prov["status"] = "rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE passes of this workload and this source"
    return d["traffic_bytes_per_launch"] * d.get("launches_per_step", 1) / max(1, launches), prov


def accuracy_block(pkg, eng, n, dev, samples=256, periodic=True):
    """The timed configuration's accelerations (tree + PM, or the tree alone for the tree-only configs) against the direct sum
    over ALL sources for `samples` seeded targets, on the GPU: the truth gravity_forcetest() uses (gravtree_forcetest.c:28-356,
    force_treeevaluate_direct forcetree.c:3428-3548 -- periodic: nearest image + lattice correction tables)."""
    import torch
    d_acc = torch.empty((n, 3), dtype=torch.float64, device=dev)
    d_pm = torch.empty((n, 3), dtype=torch.float64, device=dev) if periodic else None
    d_cost = torch.empty(n, dtype=torch.float32, device=dev)
    eng.get_accel_device(acc_ptr=d_acc.data_ptr(), pm_ptr=d_pm.data_ptr() if periodic else None, cost_ptr=d_cost.data_ptr())
    idx = np.sort(np.random.default_rng(7).choice(n, samples, replace=False)).astype(np.int32)
    sel = torch.from_numpy(idx.astype(np.int64)).to(dev)
    tot = ((d_acc[sel] + d_pm[sel]) if periodic else d_acc[sel]).cpu().numpy()
    ia = float(d_cost.double().mean().item())
    del d_acc, d_pm, d_cost
    t0 = time.time()
    truth = eng.direct_sum(idx)
    e = np.linalg.norm(tot - truth, axis=1) / np.linalg.norm(truth, axis=1)
    return {"truth": ("periodic direct sum" if periodic else "direct sum") + " over all %d sources on the GPU (ngravs_direct_sum)" % n,
            "samples": int(samples),
            "rms": float(np.sqrt(np.mean(e ** 2))), "median": float(np.median(e)), "p99": float(np.percentile(e, 99)),
            "max": float(e.max()), "ia_per_particle": ia, "seconds": time.time() - t0,
            "reference_band": ("reference TreePM walk at the same ErrTolForceAcc: rms 6.5e-3 ... 9.6e-3 (SURVEY.md 6)" if periodic else
                               "reference tree walk at the same ErrTolForceAcc on GalaxyCollision.IC / Plummer spheres: rms 3e-3 ... 6e-3 "
                               "(tests/test_gpu_parity.py)")}


def make_box(pkg, n, L, n_gravs, seed):
    rng = np.random.default_rng(seed)
    chunk = 1 << 22
    pos = np.empty((n, 3), dtype=np.float64)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        p = rng.uniform(0.0, L, (e - s, 3)).astype(np.float32)
        p[p >= L] = np.nextafter(np.float32(L), np.float32(0))
        pos[s:e] = p
    mass = np.full(n, 1.0 / n)
    ptype = (1 + (np.arange(n) % n_gravs)).astype(np.int32)
    return pos, mass, ptype


def host_cores():
    """cores this process may really use: affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
            if q != "max":
                n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    if "NGRAVS_CPU_CORES" in os.environ:
        n = int(os.environ["NGRAVS_CPU_CORES"])
    return n


def cpu_baseline(pkg, args, cells_per_particle, gpu_ia):
    """The oracle (CPU restatement of the reference algorithm) on a bounded sample of the same workload -- same density per PM
    cell / same sphere / the same IC, same wiring, same criterion; all host cores for the walk and PM (OpenMP), serial insertion
    tree build as in the reference.  Sized for 10-30 s of CPU work."""
    O = ge.load_oracle()
    ncores = host_cores()
    n_gravs, wiring = args.ngravs, args.wiring
    treeonly = args.config in ("c1", "c2")
    L = 1.0
    if args.config == "c1":
        ic = pkg.ic.read_gadget_format1(os.path.join(ROOT, "tests", "golden", "GalaxyCollision.IC"))
        pos, mass, ptype = ic["pos"], ic["mass"], ic["type"]
        n, pmgrid = len(pos), 0
        cfg = pkg.make_config(n_gravs=2, G=43007.1, theta=0.5, err_tol_force_acc=0.005, softening=[0, 1.0, 0.4, 1.0, 1.0, 1.0],
                              type_to_grav=[0, 0, 1, 0, 0, 0], wiring="newton", tree_alloc_factor=0.8)
        what = "the whole IC (%d particles)" % n
    elif args.config == "c2":
        n, pmgrid = 1 << args.log2n, 0
        pos, mass, ptype = pkg.ic.plummer_sphere(n, a=1.0, seed=12345)   # the IC the GPU ran
        cfg = pkg.make_config(n_gravs=1, periodic=0, pmgrid=0, box_size=0.0, G=1.0, theta=0.5, err_tol_force_acc=0.005,
                              softening=[0.01] * 6, type_to_grav=pkg.ic.default_type_to_grav(1), wiring="newton")
        what = "the same IC (2^%d-particle Plummer sphere)" % args.log2n
    else:
        n, pmgrid = 1 << 22, 64
        while (pmgrid * 2) ** 3 <= n * cells_per_particle:
            pmgrid *= 2
        pos, mass, ptype = make_box(pkg, n, L, n_gravs, 4242)
        eps = L / (40 * n ** (1 / 3))
        cfg = pkg.make_config(n_gravs=n_gravs, periodic=1, pmgrid=pmgrid, box_size=L, G=1.0, theta=0.5, softening=[eps] * 6,
                              type_to_grav=pkg.ic.default_type_to_grav(n_gravs), wiring=wiring)
        what = "2^22 particles, PMGRID=%d (same %g cells/particle, wiring, eps/spacing)" % (pmgrid, cells_per_particle)
    tab = O.shortrange_table(cfg)[0] if pmgrid else None
    # untimed first pass (theta) to obtain OldAcc
    dom = O.domain_extent(pos)
    T = O.Tree(cfg, pos, mass, ptype, dom)
    pm = O.pm_periodic(cfg, pos, mass, ptype) if pmgrid else None
    a, _ = T.walk(table=tab, nthreads=ncores)
    _, old = O.finish(cfg, a, pm)
    T.close()
    cfg.err_tol_theta = 0.0
    # timed steady-state step(s): decomposition + order, PM, tree build, walk
    reps = 20 if args.config == "c1" else 1    # (the 60 000-particle IC takes 0.2 s)
    tph = np.zeros(4)
    for _ in range(reps):
        t0 = time.time()
        dom = O.domain_extent(pos)
        key = O.keys(pos, dom)
        order = O.peano_order(cfg, key, ptype)
        p2, m2, t2, o2 = pos[order], mass[order], ptype[order], old[order]
        t1 = time.time()
        pm = O.pm_periodic(cfg, p2, m2, t2) if pmgrid else None
        t2_ = time.time()
        T = O.Tree(cfg, p2, m2, t2, dom)
        t3 = time.time()
        a, nint = T.walk(old_acc=o2, table=tab, nthreads=ncores)
        O.finish(cfg, a, pm)
        t4 = time.time()
        T.close()
        tph += np.array([t1 - t0, t2_ - t1, t3 - t2_, t4 - t3])
    tph /= reps
    total = float(tph.sum())
    return {
        "value": n / total, "unit": "particle-steps/s", "cores": ncores, "kind": "port",
        "sample": "%s, relative criterion; phases s: domain+order %.2f pm %.2f build %.2f walk %.2f; reference walk %.1f interactions/particle "
                  "(the GPU path's own: %.1f), %.3g interactions/s/core"
                  % (what, tph[0], tph[1], tph[2], tph[3], float(nint.mean()), gpu_ia, float(nint.sum()) / tph[3] / ncores),
    }


def self_launch(ngpus):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script with torch.distributed.run (one process per
    GPU, rendezvous on 127.0.0.1) as CHILD processes and relay what they print.  The parent initialises no GPU and no
    process group; it only waits.  Returns the children's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC only on these hosts (RCCL across processes)
    import tempfile
    with tempfile.TemporaryFile(mode="w+") as errf:
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=errf, text=True)
        for line in proc.stdout:                         # rank 0 prints the one JSON line; anything else is passed through too
            sys.stdout.write(line)
            sys.stdout.flush()
        rc = proc.wait()
        # the ranks' stderr: all of it when they failed (a communicator that gives up says which task, which collective and the
        # byte counts per peer there, and ends its process with exit code 86), its tail otherwise
        errf.seek(0)
        lines = errf.read().splitlines()
        if rc != 0:
            sys.stderr.write("bench.py: the ranks ended with exit code %d; their stderr:\n" % rc)
        for line in (lines if rc != 0 else lines[-20:]):
            sys.stderr.write(line + "\n")
        sys.stderr.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2n", type=int, default=26, help="log2 of the particle number (26 = the 64M headline)")
    ap.add_argument("--pmgrid", type=int, default=0, help="0 = 2 cells per particle (512 at 64M)")
    ap.add_argument("--ngravs", type=int, default=2)
    ap.add_argument("--wiring", default="c4")
    ap.add_argument("--walk", default="group", choices=["group", "strict"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-accuracy", action="store_true", help="skip the direct-sum accuracy check after the timed region")
    ap.add_argument("--tune", action="append", default=[], metavar="NAME=VALUE",
                    help="ngravs_set_tuning() parameters, e.g. --tune walk_sg=2 (experiments; the default run sets none)")
    ap.add_argument("--decomp", default=None, choices=["replicated", "domain"],
                    help="N>1: 'domain' (default) = work-weighted Peano-Hilbert domain decomposition: migration, all-reduced top-leaf "
                         "moments + import of the top-tree cells a task may open, x-slab decomposed PM with four plane exchanges "
                         "(DESIGN.md 7); 'replicated' = every rank holds all particles and only the walk is sharded (no data-path "
                         "collective; default for the tree-only configs)")
    ap.add_argument("--leaf-max", type=float, default=0.0,
                    help="N>1: split top-tree nodes above this many particles (0: TotNumPart/(20 NTask), at most NGRAVS_TOPLEAF_MAX = 3000); "
                         "smaller leaves = finer import granularity, larger per-leaf tables")
    ap.add_argument("--config", default="c4", choices=["c1", "c2", "c3", "c4", "c5"],
                    help="BASELINE.json config: c4 (default, the metric's) | c5: 256M N_GRAVS=3 PMGRID=1024 | c3: 16M N_GRAVS=1 PMGRID=256 | "
                         "c2: 4M Plummer tree-only | c1: the reference's own GalaxyCollision.IC (60k particles, N_GRAVS=2, tree-only)")
    args = ap.parse_args()
    if args.config == "c3":
        args.log2n, args.ngravs, args.wiring, args.pmgrid = (24 if args.log2n == 26 else args.log2n), 1, "newton", args.pmgrid or 256
    if args.config == "c5":    # 256M, N_GRAVS=3, PMGRID=1024: the 8-GPU config of BASELINE.json; it also fits ONE 288 GB MI355X
        args.log2n, args.ngravs, args.pmgrid = (28 if args.log2n == 26 else args.log2n), 3, args.pmgrid or 1024
    if args.config == "c2":
        args.log2n, args.ngravs, args.wiring = (22 if args.log2n == 26 else args.log2n), 1, "newton"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))     # before anything in this process could touch a GPU

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    # rehearsal knobs (one-GPU box): NGRAVS_BENCH_BACKEND=gloo NGRAVS_BENCH_DEVICE=0 puts every rank on GPU 0
    backend = os.environ.get("NGRAVS_BENCH_BACKEND", "nccl")
    if "NGRAVS_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["NGRAVS_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    pkg = ge.load_package()
    if not os.path.exists(pkg.LIB_PATH):
        if rank == 0:
            ge.build()
        if world > 1:
            dist.barrier()

    n = 1 << args.log2n
    L = 1.0
    pmgrid = args.pmgrid
    if pmgrid == 0:
        pmgrid = 16
        while (pmgrid * 2) ** 3 <= 2 * n:      # largest power of two with <= 2 cells per particle (512 at 2^26)
            pmgrid *= 2
    cells_per_particle = pmgrid ** 3 / n
    eps = L / (40 * n ** (1 / 3))
    treeonly = args.config in ("c1", "c2")
    if treeonly:
        pmgrid, cells_per_particle, eps = 0, 0.0, 0.01
    if args.config == "c1":
        # the reference's shipped IC with the parameters of its Configuration.reference
        ic = pkg.ic.read_gadget_format1(os.path.join(ROOT, "tests", "golden", "GalaxyCollision.IC"))
        pos, mass, ptype = ic["pos"], ic["mass"], ic["type"]
        n, args.ngravs, args.wiring = len(pos), 2, "newton"
        cfg = pkg.make_config(n_gravs=2, G=43007.1, theta=0.5, err_tol_force_acc=0.005, softening=[0, 1.0, 0.4, 1.0, 1.0, 1.0],
                              type_to_grav=[0, 0, 1, 0, 0, 0], wiring="newton", tree_alloc_factor=0.8,
                              walk_mode=pkg.WALK_GROUP if args.walk == "group" else pkg.WALK_STRICT,
                              device=local_rank, rank=rank, world_size=world)
    else:
        cfg = pkg.make_config(n_gravs=args.ngravs, periodic=0 if treeonly else 1, pmgrid=pmgrid, box_size=0.0 if treeonly else L,
                              G=1.0, theta=0.5, err_tol_force_acc=0.005, softening=[eps] * 6,
                              type_to_grav=pkg.ic.default_type_to_grav(args.ngravs), wiring=args.wiring,
                              walk_mode=pkg.WALK_GROUP if args.walk == "group" else pkg.WALK_STRICT,
                              device=local_rank, rank=rank, world_size=world)
        if treeonly:
            pos, mass, ptype = pkg.ic.plummer_sphere(n, a=1.0, seed=12345)
        else:
            pos, mass, ptype = make_box(pkg, n, L, args.ngravs, 12345)
    dev = torch.device("cuda", local_rank)
    d_pos = torch.from_numpy(pos).to(dev)
    d_mass = torch.from_numpy(mass).to(dev)
    d_type = torch.from_numpy(ptype).to(dev)
    d_old = torch.zeros(n, dtype=torch.float64, device=dev)
    del pos, mass, ptype
    if args.decomp is None:
        args.decomp = "domain" if world > 1 and not treeonly else "replicated"
    domain = args.decomp == "domain" and world > 1
    if domain:
        import importlib
        dd = importlib.import_module("ngravs_amd.distributed")
        eng = dd.DistributedEngine(cfg, leaf_max=args.leaf_max or None, comm=os.environ.get("NGRAVS_BENCH_COMM") or None)
        sel = torch.arange(rank, n, world, device=dev)          # arbitrary initial ownership; the first step migrates
        l_pos, l_mass, l_type = d_pos[sel].contiguous(), d_mass[sel].contiguous(), d_type[sel].contiguous()
        eng.set_particles_device(int(sel.numel()), l_pos.data_ptr(), l_mass.data_ptr(), l_type.data_ptr())
        eng._check(pkg.lib().ngravs_dd_set_ids(eng._h, sel.data_ptr(), 1), "ngravs_dd_set_ids")
        del d_pos, d_mass, d_type
    else:
        eng = pkg.Engine(cfg)
        eng.set_particles_device(n, d_pos.data_ptr(), d_mass.data_ptr(), d_type.data_ptr())
    for kv in args.tune:
        k, v = kv.split("=")
        eng.set_tuning(**{k: float(v)})
    torch.cuda.synchronize()

    # pass 1 (untimed): Barnes-Hut theta=0.5 with OldAcc=0, as the reference's first force computation
    eng.compute_accelerations(pm_step=True)
    if world > 1:
        # every rank needs OldAcc of all particles only for ITS targets; its own shard is what it has
        pass
    if domain:
        # OldAcc of the own particles (first num_local rows of the working set) feeds the next pass
        nl = eng.num_local()
        tmp = torch.zeros(max(nl, 1), dtype=torch.float64, device=dev)   # own rows: what the library delivers and what it reads back
        eng.get_old_acc_device(tmp.data_ptr())
        eng._check(pkg.lib().ngravs_set_old_acc(eng._h, tmp.data_ptr(), 8, 1), "ngravs_set_old_acc")
    else:
        eng.get_old_acc_device(d_old.data_ptr())
        eng.set_old_acc_device(d_old.data_ptr())
    eng.set_opening(0.0, 0.005)     # All.ErrTolTheta = 0 latch (gravtree.c:334-335)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.compute_accelerations(pm_step=True)
    walk_ms, phases, eval_ms = [], [], []
    if domain:
        eng.reset_wall()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.compute_accelerations(pm_step=True)
        st = eng.stats()
        walk_ms.append(st.walk_kernel_ms)
        eval_ms.append((st.reserved[4], st.reserved[5], st.reserved[6]))
        phases.append((st.t_domain + st.t_peano, st.t_pm, st.t_treebuild, st.t_treewalk))
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    st = eng.stats()
    shard_first, shard_count = eng.shard()
    if domain:
        shard_count = eng.num_local()

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = n * args.steps / dt
        k_ms = float(np.mean(walk_ms))
        alg = walk_alg_bytes(args.ngravs) * shard_count
        kname = "k_walk_group2<..,0> (fused)" if args.walk == "group" else "k_walk_strict"
        ev = np.mean(np.array(eval_ms), axis=0)
        split = None
        if args.walk == "group" and ev[1] > 0:
            # split walk: the dominant kernel is the evaluation kernel, launched once per batch of groups; one launch
            # processes shard_count/launches targets on average (HIP events around every launch, inside the library)
            split = {"launches_per_step": int(ev[1]), "eval_ms_per_step": float(ev[0]), "traversal_ms_per_step": float(ev[2])}
            ring = (not treeonly) and not any(kv.replace(" ", "") in ("walk_ring=0", "walk_ring=0.0") for kv in args.tune) and \
                (args.ngravs <= 2 or True)
            kname = "k_eval_ring (evaluation, ring pool)" if ring else "k_walk_group2<..,2> (evaluation)"
            k_ms = float(ev[0] / ev[1])
            alg = alg / ev[1]
        achieved = alg / (k_ms * 1e-3) / 1e9
        ph = np.mean(np.array(phases), axis=0)
        out = {
            "metric": "particle-steps/s (tree+PM)", "value": value, "unit": "particle-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d-particle %s, N_GRAVS=%d (%s wiring), %s, relative criterion ErrTolForceAcc=0.005, %s walk" %
                                   (args.config.upper(), n, "GalaxyCollision.IC" if args.config == "c1" else ("Plummer sphere" if treeonly else "uniform periodic box"), args.ngravs,
                                    args.wiring, "tree-only" if treeonly else "TreePM PMGRID=%d" % pmgrid, args.walk),
                       "particles": n, "n_gravs": args.ngravs, "pmgrid": pmgrid, "walk": args.walk,
                       "mesh_cells_per_particle": cells_per_particle,   # the short-range sphere holds ~ 1/this: ia_per_particle scales with it
                       "backend": ((eng.backend + ("; " + eng.comm_note if getattr(eng, "comm_note", "") else "")) if domain
                                   else (dist.get_backend() if world > 1 else None)),
                       "world_size_reported_by_backend": (eng.comm.world_reported if domain else (dist.get_world_size() if world > 1 else 1)),
                       "parallelism": ("work-weighted Peano-Hilbert domain decomposition over %d tasks: migration + halo all-to-all-v, x-slab decomposed PM (4 plane exchanges)" % world)
                       if domain else ("walk sharded over %d Peano segment(s); decomposition, build, PM replicated" % world),
                       "untimed_prologue": ("every rank generates the whole seeded IC on its host (%d particles: the ranks must agree on it without a file), "
                                            "uploads it and keeps every %d-th particle; the first step migrates" % (n, world)) if domain else
                                           "the IC is generated on the host and uploaded once",
                       "phases_ms": {"domain+peano": ph[0] * 1e3, "pm": ph[1] * 1e3, "treebuild": ph[2] * 1e3,
                                     "treewalk": ph[3] * 1e3},
                       "ia_per_particle": st.interactions / max(1, st.n_active), "tree_nodes": st.n_nodes,
                       "walk_list_entries_per_group": st.reserved[0], "walk_nodes_tested_per_group": st.reserved[1],
                       "walk_batches_per_group": st.reserved[2], "walk_force_iters_per_group": st.reserved[3],
                       "step_algorithmic_bytes_per_particle": step_alg_bytes(args.ngravs, cells_per_particle, pm=not treeonly),
                       "step_fraction_of_hbm_roofline": value / world * step_alg_bytes(args.ngravs, cells_per_particle, pm=not treeonly) / HBM_PEAK},
            "roofline": {"bound": "hbm", "kernel": kname, "split_walk": split,
                         "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / (HBM_PEAK / 1e9), "traffic": None,
                         "algorithmic_bytes_per_launch": alg, "avg_launch_ms": k_ms},
        }
        out["roofline"]["traffic"], out["roofline"]["traffic_provenance"] = walk_traffic(args, n, world, split["launches_per_step"] if split else 1)
        if args.walk == "group" and split:
            # the kernel's own bound: fp64 VALU issue.  Every evaluated pair costs one trip of the force loop's common path (static
            # count from this build's assembly, profiles/r04_eval_isa_mix.txt, ER_TRIP4_ASM: 44.5 VALU (counted as 45) + 3 extra issue slots for the quarter-rate
            # v_rsq_f64), one wave-instruction issues in 4 cycles on a SIMD for 64
            # lanes, 4 SIMDs per CU
            pairs = st.interactions / max(1, split["launches_per_step"])
            # (tree-only walks with one lane per target -- force-loop trips per group = pairs per particle -- run ER_DIRECT_ASM: 21 + 3;
            # with several lanes per target or per-pair images k_walk_group2's C++ loop: 49 + 3)
            ia_pp = st.interactions / max(1, st.n_active)
            direct_asm = treeonly and ia_pp > 0 and abs(st.reserved[3] / ia_pp - 1.0) < 0.01
            slots, clock_hz, simds = (48 if ring else (24 if direct_asm else 52)), 2.4e9, 4 * torch.cuda.get_device_properties(dev).multi_processor_count
            floor_ms = pairs * slots * 4.0 / 64.0 / (simds * clock_hz) * 1e3
            out["roofline"]["secondary"] = {"bound": "fp64 VALU issue", "floor_ms": floor_ms, "achieved_ms": k_ms, "frac": floor_ms / k_ms,
                                            "issue_slots_per_pair": slots, "pairs_per_launch": pairs, "simds": simds, "clock_ghz": clock_hz / 1e9,
                                            "note": "perfectly packed lanes, nothing but the force loop; the kernel also fetches, culls and "
                                                    "masks its lists, and its lanes finish their lists at different times (DESIGN.md 5)"}
        if domain:
            # rank 0's host wall clock per step: the three stages of compute_accelerations() and the part of each spent inside
            # collectives (waiting for the slowest task included); payloads of the last step
            w = eng.wall
            k = max(1, w["steps"])
            out["config"]["rank0_wall_ms_per_step"] = {
                "decomposition": w["decomposition_s"] / k * 1e3, "decomposition_in_collectives": w["decomposition_collectives_s"] / k * 1e3,
                "pm": w["pm_s"] / k * 1e3, "pm_in_collectives": w["pm_collectives_s"] / k * 1e3,
                "gravity_tree": w["gravity_tree_s"] / k * 1e3, "collective_calls": w["collective_calls"] / k}
            out["config"]["rank0_last_step"] = {
                "n_local": int(eng.info.n_local), "n_imported": int(eng.info.n_halo), "n_migrated_in": int(eng.info.n_migrated_in),
                "work_balance": float(eng.info.work_balance), "memory_balance": float(eng.info.memory_balance),
                "bytes_migration": float(eng.info.bytes_migration), "bytes_import": float(eng.info.bytes_halo),
                "pm_exchange_bytes": eng.pm_bytes() if not treeonly else None,
                "pm_stage_ms": (lambda v: {"deposit+boxes": v[0], "pack": [v[1], v[4], v[7], v[10]], "alltoallv": [v[2], v[5], v[8], v[11]],
                                           "unpack": [v[3], v[6], v[9], v[12]]})([1e3 * x for x in eng.pm_seconds()]) if not treeonly else None,
                "top_tree": {"nodes": int(eng.info.n_topnodes), "leaves": int(eng.info.n_topleaves), "counting_rounds": int(eng.info.toptree_rounds)},
                "decomposition_collectives": int(eng.info.collectives),
                "decomposition_stage_ms": dict(zip(DD_STAGES, [1e3 * float(v) for v in eng.info.seconds]))}
        if world == 1 and not args.no_accuracy:
            out["accuracy"] = accuracy_block(pkg, eng, n, dev, periodic=not treeonly)
        else:
            out["accuracy"] = None
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pkg, args, cells_per_particle, st.interactions / max(1, st.n_active))
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
