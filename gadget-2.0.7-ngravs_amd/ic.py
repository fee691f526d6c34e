"""Initial conditions for the force path: the synthetic boxes of SURVEY.md 8(d) and a minimal
Gadget snapshot-format-1 reader (header + POS + ID + optional MASS; SURVEY.md Appendix E,
reference read_ic.c:244-612, allvars.h:685-708) for config C1 (GalaxyCollision.IC).
"""
import struct

import numpy as np


def uniform_box(n, box=1.0, n_gravs=1, seed=12345, dtype32=True):
    """iid U[0,L)^3, cast to fp32 then widened (Gadget ICs are fp32 on disk), equal masses 1/n,
    species i mod n_gravs carried by particle types 1..n_gravs (GravityHalo=0, GravityDisk=1, ...)."""
    rng = np.random.default_rng(seed)
    pos = rng.uniform(0.0, box, (n, 3))
    if dtype32:
        pos = pos.astype(np.float32).astype(np.float64)
        pos[pos >= box] = np.nextafter(np.float32(box), np.float32(0)).astype(np.float64)
    mass = np.full(n, 1.0 / n)
    ptype = (1 + (np.arange(n) % n_gravs)).astype(np.int32)
    return pos, mass, ptype


def plummer_sphere(n, a=1.0, seed=12345, rmax=100.0):
    """Plummer sphere, total mass 1, radii by inverse CDF r=a/sqrt(u^(-2/3)-1) truncated at rmax*a."""
    rng = np.random.default_rng(seed)
    r = np.empty(0)
    while len(r) < n:
        u = rng.uniform(0.0, 1.0, n)
        rr = a / np.sqrt(u ** (-2.0 / 3.0) - 1.0)
        r = np.concatenate([r, rr[rr < rmax * a]])
    r = r[:n]
    ct = rng.uniform(-1.0, 1.0, n)
    ph = rng.uniform(0.0, 2 * np.pi, n)
    st = np.sqrt(1 - ct * ct)
    pos = np.stack([r * st * np.cos(ph), r * st * np.sin(ph), r * ct], axis=1)
    pos = pos.astype(np.float32).astype(np.float64)
    mass = np.full(n, 1.0 / n)
    ptype = np.ones(n, dtype=np.int32)
    return pos, mass, ptype


def default_type_to_grav(n_gravs):
    """types 1..n_gravs -> species 0..n_gravs-1, everything else species 0"""
    t2g = [0] * 6
    for g in range(n_gravs):
        t2g[1 + g] = g
    return t2g


def read_gadget_format1(path):
    """Returns dict(pos[N,3] f64, vel, ids, mass[N] f64, type[N] i32, header)."""
    with open(path, "rb") as f:
        raw = f.read()
    off = 0

    def block():
        nonlocal off
        (nb,) = struct.unpack_from("<i", raw, off)
        payload = raw[off + 4: off + 4 + nb]
        (nb2,) = struct.unpack_from("<i", raw, off + 4 + nb)
        if nb != nb2:
            raise ValueError("corrupt record at %d" % off)
        off += 8 + nb
        return payload

    h = block()
    npart = np.frombuffer(h, dtype="<i4", count=6, offset=0).astype(np.int64)
    masstab = np.frombuffer(h, dtype="<f8", count=6, offset=24)
    time, redshift = struct.unpack_from("<dd", h, 72)
    boxsize = struct.unpack_from("<d", h, 72 + 16 + 8 + 24 + 8)[0]
    n = int(npart.sum())
    pos = np.frombuffer(block(), dtype="<f4", count=3 * n).reshape(n, 3).astype(np.float64)
    vel = np.frombuffer(block(), dtype="<f4", count=3 * n).reshape(n, 3).astype(np.float64)
    ids = np.frombuffer(block(), dtype="<u4", count=n).copy()
    ptype = np.repeat(np.arange(6, dtype=np.int32), npart)
    mass = np.repeat(masstab, npart).astype(np.float64)
    nwithmass = int(sum(npart[t] for t in range(6) if masstab[t] == 0 and npart[t] > 0))
    if nwithmass > 0:
        mblk = np.frombuffer(block(), dtype="<f4", count=nwithmass).astype(np.float64)
        k = 0
        start = 0
        for t in range(6):
            if masstab[t] == 0 and npart[t] > 0:
                mass[start:start + npart[t]] = mblk[k:k + npart[t]]
                k += npart[t]
            start += npart[t]
    return dict(pos=pos, vel=vel, ids=ids, mass=mass, type=ptype,
                header=dict(npart=npart, mass=masstab, time=time, redshift=redshift, boxsize=boxsize))


def write_gadget_format1(path, pos, vel, ids, ptype, masstab, mass=None, time=0.0, boxsize=0.0):
    """Snapshot/IC format 1 (reference io.c:672-996, header allvars.h:685-708): header, POS, VEL,
    ID and a MASS block for the types whose header mass is 0.  Particles are written grouped by type, fp32 on disk."""
    pos, vel = np.asarray(pos, dtype=np.float64), np.asarray(vel, dtype=np.float64)
    ids, ptype = np.asarray(ids, dtype=np.uint32), np.asarray(ptype, dtype=np.int32)
    masstab = np.asarray(masstab, dtype=np.float64)
    order = np.argsort(ptype, kind="stable")
    npart = np.bincount(ptype, minlength=6).astype(np.int32)
    hdr = bytearray(256)
    struct.pack_into("<6i", hdr, 0, *npart)
    struct.pack_into("<6d", hdr, 24, *masstab)
    struct.pack_into("<dd", hdr, 72, time, 0.0)
    struct.pack_into("<ii", hdr, 88, 0, 0)
    struct.pack_into("<6I", hdr, 96, *npart)
    struct.pack_into("<ii", hdr, 120, 0, 1)
    struct.pack_into("<d", hdr, 128, boxsize)

    def rec(f, payload):
        f.write(struct.pack("<i", len(payload)))
        f.write(payload)
        f.write(struct.pack("<i", len(payload)))

    with open(path, "wb") as f:
        rec(f, bytes(hdr))
        rec(f, pos[order].astype("<f4").tobytes())
        rec(f, vel[order].astype("<f4").tobytes())
        rec(f, ids[order].astype("<u4").tobytes())
        need = [t for t in range(6) if masstab[t] == 0 and npart[t] > 0]
        if need:
            if mass is None:
                raise ValueError("per-particle masses needed for types %s" % need)
            m = np.asarray(mass, dtype=np.float64)[order]
            t_sorted = ptype[order]
            rec(f, np.concatenate([m[t_sorted == t] for t in need]).astype("<f4").tobytes())
