"""ctypes mirror of include/ngravs_hip.h (plain-C ABI types shared by the engine and the tests).

Field order/types must match the header exactly; tests/test_abi.py checks sizeof against the
compiled library (ngravs_build_info) so that a drift is caught on CPU.
"""
import ctypes as C

ABI_VERSION = 3
MAX_GRAVS = 3
NTYPES = 6
NTAB = 2048
ASMTH = 1.25
RCUT = 4.5
BITS_PER_DIMENSION = 18

LAW_NONE, LAW_NEWTON, LAW_NEG_NEWTON, LAW_YUKAWA, LAW_COLOYUK, LAW_BAMBAM, LAW_SOURCEBAM, LAW_TARGETBAM = range(8)
SPLINE_NONE, SPLINE_PLUMMER, SPLINE_NEG_PLUMMER, SPLINE_BAMBAM, SPLINE_SOURCEBAM, SPLINE_TARGETBAM = range(6)
WALK_STRICT, WALK_GROUP = 0, 1

LAW_NAMES = {"none": LAW_NONE, "newtonian": LAW_NEWTON, "neg_newtonian": LAW_NEG_NEWTON,
             "yukawa": LAW_YUKAWA, "coloyuk": LAW_COLOYUK}
SPLINE_NAMES = {"none": SPLINE_NONE, "plummer": SPLINE_PLUMMER, "neg_plummer": SPLINE_NEG_PLUMMER}

_G3 = (C.c_int32 * MAX_GRAVS) * MAX_GRAVS


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("n_gravs", C.c_int32), ("periodic", C.c_int32), ("pmgrid", C.c_int32),
        ("box_size", C.c_double), ("G", C.c_double), ("err_tol_theta", C.c_double),
        ("err_tol_force_acc", C.c_double),
        ("force_softening", C.c_double * NTYPES), ("type_to_grav", C.c_int32 * NTYPES),
        ("law_accel", _G3), ("law_spline", _G3), ("law_greens", _G3), ("law_normed", _G3),
        ("yukawa_imass", C.c_double), ("asmth", C.c_double), ("rcut", C.c_double),
        ("tree_alloc_factor", C.c_double), ("group_reach", C.c_double),
        ("walk_mode", C.c_int32), ("device", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32),
        ("bam_epsilon", C.c_double), ("reserved", C.c_int32 * 6),
    ]


class Particles(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("pos", C.c_void_p), ("pos_stride", C.c_int64),
        ("mass", C.c_void_p), ("mass_stride", C.c_int64),
        ("type", C.c_void_p), ("type_stride", C.c_int64),
        ("old_acc", C.c_void_p), ("old_acc_stride", C.c_int64),
        ("active", C.c_void_p), ("active_stride", C.c_int64),
        ("grav_pm", C.c_void_p), ("grav_pm_stride", C.c_int64),
        ("grav_cost", C.c_void_p), ("grav_cost_stride", C.c_int64),
        ("on_device", C.c_int32), ("reserved", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("n_active", C.c_int64), ("n_nodes", C.c_int64), ("interactions", C.c_double),
        ("t_domain", C.c_double), ("t_peano", C.c_double), ("t_treebuild", C.c_double),
        ("t_treewalk", C.c_double), ("t_pm", C.c_double), ("walk_kernel_ms", C.c_double),
        ("reserved", C.c_double * 7),
    ]


def make_config(n_gravs=1, periodic=0, pmgrid=0, box_size=0.0, G=1.0, theta=0.5, err_tol_force_acc=0.005,
                softening=None, type_to_grav=None, wiring="newton", yukawa_imass=60.0, walk_mode=WALK_STRICT,
                tree_alloc_factor=0.0, device=0, rank=0, world_size=1, group_reach=0.0):
    """Build a Config the way init_grav_maps()+wire_grav_maps() would (ngravs_core.c:201, ngravs.c:64).

    softening: Plummer-equivalent eps per type (SofteningTable); ForceSoftening = 2.8*eps (gravtree.c:514).
    wiring: 'newton'    all pairs Newtonian            (NGRAVS_STOCK_TESTING, ngravs.c:98-146)
            'coloyuk'   all pairs Newton+Yukawa        (NGRAVS_COMBINED_TESTING_UNIFORM, ngravs.c:284-320)
            'yukawa_offdiag' diagonal none, off-diagonal Yukawa (NGRAVS_YUKAWA_FORCETEST, ngravs.c:213-283)
            'c4'        diagonal Newton, off-diagonal Newton+Yukawa (SURVEY.md 8(d) research wiring for C4/C5)
            'bam'       N_GRAVS=2: species 0 baryons, species 1 BAM (NGRAVS_ACCUMULATOR_TESTING, ngravs.c:163-210): [0][0]
                        newtonian/plummer, [0][1] sourcebambaryon, [1][0] sourcebaryonbam, [1][1] bambam; tree-only
    """
    cfg = Config()
    cfg.abi_version = ABI_VERSION
    cfg.n_gravs = n_gravs
    cfg.periodic = int(periodic)
    cfg.pmgrid = int(pmgrid)
    cfg.box_size = float(box_size)
    cfg.G = float(G)
    cfg.err_tol_theta = float(theta)
    cfg.err_tol_force_acc = float(err_tol_force_acc)
    softening = softening if softening is not None else [0.0] * NTYPES
    for t in range(NTYPES):
        cfg.force_softening[t] = 2.8 * float(softening[t])
    type_to_grav = type_to_grav if type_to_grav is not None else [0] * NTYPES
    for t in range(NTYPES):
        cfg.type_to_grav[t] = int(type_to_grav[t])
    for i in range(n_gravs):
        for j in range(n_gravs):
            if wiring == "newton":
                law, spl = LAW_NEWTON, SPLINE_PLUMMER
            elif wiring == "coloyuk":
                law, spl = LAW_COLOYUK, SPLINE_PLUMMER
            elif wiring == "yukawa_offdiag":
                law, spl = (LAW_NONE, SPLINE_NONE) if i == j else (LAW_YUKAWA, SPLINE_PLUMMER)
            elif wiring == "c4":
                law, spl = (LAW_NEWTON if i == j else LAW_COLOYUK), SPLINE_PLUMMER
            elif wiring == "bam":
                law, spl = {(0, 0): (LAW_NEWTON, SPLINE_PLUMMER), (0, 1): (LAW_SOURCEBAM, SPLINE_SOURCEBAM),
                            (1, 0): (LAW_TARGETBAM, SPLINE_TARGETBAM), (1, 1): (LAW_BAMBAM, SPLINE_BAMBAM)}[(i, j)]
            else:
                raise ValueError("unknown wiring %r" % wiring)
            cfg.law_accel[i][j] = law
            cfg.law_spline[i][j] = spl
            cfg.law_greens[i][j] = law if wiring != "bam" else LAW_NONE
            cfg.law_normed[i][j] = law if wiring != "bam" else LAW_NONE
    cfg.yukawa_imass = float(yukawa_imass)
    cfg.tree_alloc_factor = float(tree_alloc_factor)
    cfg.group_reach = float(group_reach)
    cfg.walk_mode = int(walk_mode)
    cfg.device = int(device)
    cfg.rank = int(rank)
    cfg.world_size = int(world_size)
    return cfg
