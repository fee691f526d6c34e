import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
pkg = ge.load_package()
n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18, 1.0
pm = int(sys.argv[2]) if len(sys.argv) > 2 else 64
pos, mass, typ = pkg.ic.uniform_box(n, box=L, n_gravs=1, seed=11)
eps = L / (40 * n ** (1 / 3))
res = {}
for name, tune in (("old", {"walk_ring": 0}), ("ring", {}), ("ring_exact", {"walk_exact_reach": 1}), ("ring_k4", {"walk_ring_k": 4})):
    cfg = pkg.make_config(n_gravs=1, periodic=1, pmgrid=pm, box_size=L, G=1.0, theta=0.5, softening=[eps] * 6,
                          type_to_grav=pkg.ic.default_type_to_grav(1), wiring="newton", walk_mode=pkg.WALK_GROUP)
    eng = pkg.Engine(cfg)
    eng.set_tuning(**tune)
    eng.set_particles(pos, mass, typ)
    eng.compute_accelerations(pm_step=True)
    a, o, c = eng.get_accel()
    eng.set_old_acc(o)
    eng.set_opening(0.0, 0.005)
    eng.compute_accelerations(pm_step=True)
    a, o, c = eng.get_accel()
    st = eng.stats()
    print(name, "entries/group %.1f trips %.1f" % (st.reserved[0], st.reserved[3]))
    res[name] = (a, c)
    eng.close()
a0, c0 = res["old"]
for k, (a, c) in res.items():
    e = np.linalg.norm(a - a0, axis=1) / np.linalg.norm(a0, axis=1)
    print(k, "rel diff median %.2e p99 %.2e max %.2e; cost mean %.3f (old %.3f); rows with cost diff %d; nan %d" % (np.median(e), np.quantile(e, 0.99), e.max(), c.mean(), c0.mean(), (c != c0).sum(), np.isnan(a).sum()))
    if k == "ring":
        bad = np.argsort(-e)[:5]
        print("  worst rows", bad, e[bad], c[bad], c0[bad])
        print("  ratio |a|/|a0| median", np.median(np.linalg.norm(a, axis=1) / np.linalg.norm(a0, axis=1)))
