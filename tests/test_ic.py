"""format-1 reader/writer and the synthetic ICs of SURVEY.md 8(d)."""
import os

import numpy as np

from conftest import galaxy_ic


def test_galaxy_collision_header(pkg):
    d = galaxy_ic(pkg)
    assert list(d["header"]["npart"]) == [0, 10000, 20000, 10000, 10000, 10000]
    assert abs(d["header"]["mass"][1] - 0.00104634) < 1e-8 and abs(d["header"]["mass"][2] - 0.00023252) < 1e-8
    assert d["pos"].shape == (60000, 3) and len(np.unique(d["ids"])) == 60000


def test_format1_writer_reproduces_the_references_ic_byte_for_byte(pkg, tmp_path):
    """The reference's own shipped IC (tests/golden/GalaxyCollision.IC, format 1: io.c:672-996, read_ic.c:244-612), read and
    written again, is the same file -- header, record markers, POS / VEL / ID blocks, no MASS block (all masses in the table)."""
    src = os.path.join(os.path.dirname(__file__), "golden", "GalaxyCollision.IC")
    d = pkg.ic.read_gadget_format1(src)
    out = os.path.join(str(tmp_path), "again.ic")
    pkg.ic.write_gadget_format1(out, d["pos"], d["vel"], d["ids"], d["type"], d["header"]["mass"], mass=d["mass"],
                                time=d["header"]["time"], boxsize=d["header"]["boxsize"], num_files=d["header"]["num_files"])
    assert open(out, "rb").read() == open(src, "rb").read()


def test_format1_roundtrip(pkg, tmp_path):
    rng = np.random.default_rng(0)
    n = 1000
    ptype = rng.integers(1, 4, n).astype(np.int32)
    pos = rng.uniform(0, 10, (n, 3)).astype(np.float32).astype(np.float64)
    vel = rng.normal(0, 1, (n, 3)).astype(np.float32).astype(np.float64)
    ids = np.arange(n, dtype=np.uint32)
    mass = rng.uniform(1, 2, n).astype(np.float32).astype(np.float64)
    masstab = [0, 0.5, 0, 0.25, 0, 0]              # type 2 carries per-particle masses
    mass[ptype == 1] = 0.5
    mass[ptype == 3] = 0.25
    path = os.path.join(str(tmp_path), "snap.ic")
    pkg.ic.write_gadget_format1(path, pos, vel, ids, ptype, masstab, mass=mass, boxsize=10.0)
    d = pkg.ic.read_gadget_format1(path)
    order = np.argsort(ptype, kind="stable")
    assert np.array_equal(d["pos"], pos[order]) and np.array_equal(d["vel"], vel[order])
    assert np.array_equal(d["ids"], ids[order]) and np.array_equal(d["type"], ptype[order])
    assert np.allclose(d["mass"], mass[order]) and d["header"]["boxsize"] == 10.0


def test_synthetic_ics(pkg):
    pos, mass, typ = pkg.ic.uniform_box(10000, box=3.0, n_gravs=3, seed=1)
    assert pos.min() >= 0 and pos.max() < 3.0 and abs(mass.sum() - 1) < 1e-12 and set(typ) == {1, 2, 3}
    pos, mass, typ = pkg.ic.plummer_sphere(20000, seed=2)
    r = np.linalg.norm(pos, axis=1)
    assert r.max() < 100.0 and abs(np.median(r) - 1.305) < 0.05      # Plummer half-mass radius = 1.305 a


def test_format2_multifile_and_gas_roundtrip(pkg, tmp_path):
    """format 2 (labelled blocks), multi-file sets and the U block of gas particles: what one format writes the reader gets
    back grouped by type (read_ic.c order), whatever the number of files; format is detected from the first record"""
    rng = np.random.default_rng(1)
    n = 2000
    ptype = rng.integers(0, 4, n).astype(np.int32)
    pos = rng.uniform(0, 10, (n, 3)).astype(np.float32).astype(np.float64)
    vel = rng.normal(0, 1, (n, 3)).astype(np.float32).astype(np.float64)
    ids = rng.permutation(n).astype(np.uint32)
    mass = rng.uniform(1, 2, n).astype(np.float32).astype(np.float64)
    masstab = [0, 0.5, 0, 0.25, 0, 0]
    mass[ptype == 1], mass[ptype == 3] = 0.5, 0.25
    u = rng.uniform(0.1, 1, n).astype(np.float32).astype(np.float64)
    order = np.argsort(ptype, kind="stable")
    for fmt, nf in ((1, 1), (2, 1), (1, 3), (2, 4)):
        base = os.path.join(str(tmp_path), "snap_%d_%d" % (fmt, nf))
        paths = pkg.ic.write_snapshot(base, pos, vel, ids, ptype, masstab, mass=mass, u=u, boxsize=10.0, snap_format=fmt,
                                      num_files=nf, time=0.5)
        assert len(paths) == nf
        d = pkg.ic.read_snapshot(base)
        assert d["header"]["format"] == fmt and d["header"]["num_files"] == nf and d["header"]["time"] == 0.5
        assert list(d["header"]["npart_total"]) == list(np.bincount(ptype, minlength=6))
        assert np.array_equal(d["type"], ptype[order])
        # within a type the files are concatenated in order, so the stable by-type order is reproduced
        assert np.array_equal(d["ids"], ids[order]) and np.array_equal(d["pos"], pos[order]) and np.array_equal(d["vel"], vel[order])
        assert np.allclose(d["mass"], mass[order]) and np.allclose(d["u"], u[order][: (ptype == 0).sum()])
    # the format-1 reader of C1 and the general reader agree on the shipped IC
    from conftest import galaxy_ic
    g1 = galaxy_ic(pkg)
    import glob
    cand = glob.glob(os.path.join(os.path.dirname(__file__), "golden", "GalaxyCollision*"))
    g2 = pkg.ic.read_snapshot(cand[0])
    assert np.array_equal(g1["pos"], g2["pos"]) and np.array_equal(g1["ids"], g2["ids"]) and np.array_equal(g1["mass"], g2["mass"])
