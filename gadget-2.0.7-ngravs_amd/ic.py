"""Initial conditions for the force path: the synthetic boxes of SURVEY.md 8(d) and the Gadget snapshot / IC
file formats 1 and 2 (header, POS, VEL, ID, MASS, U; single files and multi-file sets; SURVEY.md Appendix E, reference
read_ic.c:244-612, io.c:672-996, allvars.h:685-708); config C1 is the shipped GalaxyCollision.IC (format 1).
"""
import os
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
    num_files = struct.unpack_from("<i", h, 124)[0]
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
                header=dict(npart=npart, mass=masstab, time=time, redshift=redshift, boxsize=boxsize, num_files=num_files))


def write_gadget_format1(path, pos, vel, ids, ptype, masstab, mass=None, time=0.0, boxsize=0.0, num_files=1):
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
    struct.pack_into("<ii", hdr, 120, 0, int(num_files))
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


# ---------------------------------------------------------------------------------------------------------------
# Snapshot formats 1 and 2, multi-file sets, gas internal energy (reference io.c:672-996, read_ic.c:244-612).
# Format 2 precedes every block by a small record {4-char label, int32 nextblock} (io.c:796-803, 832-839);
# a snapshot written by several tasks is a set base.0 .. base.(num_files-1), each holding header.npart particles of
# header.npartTotal (read_ic.c:323-333).
# ---------------------------------------------------------------------------------------------------------------
_LABELS = {"HEAD": "HEAD", "POS": "POS ", "VEL": "VEL ", "ID": "ID  ", "MASS": "MASS", "U": "U   "}


def _header_bytes(npart, masstab, time, boxsize, npart_total, num_files):
    hdr = bytearray(256)
    struct.pack_into("<6i", hdr, 0, *[int(x) for x in npart])
    struct.pack_into("<6d", hdr, 24, *[float(x) for x in masstab])
    struct.pack_into("<dd", hdr, 72, time, 0.0)
    struct.pack_into("<ii", hdr, 88, 0, 0)
    struct.pack_into("<6I", hdr, 96, *[int(x) & 0xFFFFFFFF for x in npart_total])
    struct.pack_into("<ii", hdr, 120, 0, int(num_files))
    struct.pack_into("<d", hdr, 128, boxsize)
    struct.pack_into("<6I", hdr, 176, *[int(x) >> 32 for x in npart_total])      # npartTotalHighWord (allvars.h:701)
    return bytes(hdr)


def write_snapshot(base, pos, vel, ids, ptype, masstab, mass=None, u=None, time=0.0, boxsize=0.0, snap_format=1, num_files=1):
    """Snapshot in format 1 or 2 (All.SnapFormat), optionally split over num_files files base.0 ... (every file holds a
    contiguous share of each type, as the write tasks of io.c do).  fp64 input is down-cast to fp32 on disk (the shipped
    build has no OUTPUT_IN_DOUBLEPRECISION).  `u`: internal energy per gas particle (type 0), block U."""
    if snap_format not in (1, 2):
        raise ValueError("snap_format must be 1 or 2 (HDF5, format 3, is not built)")
    pos, vel = np.asarray(pos, dtype=np.float64), np.asarray(vel, dtype=np.float64)
    ids, ptype = np.asarray(ids, dtype=np.uint32), np.asarray(ptype, dtype=np.int32)
    masstab = np.asarray(masstab, dtype=np.float64)
    order = np.argsort(ptype, kind="stable")
    tot = np.bincount(ptype, minlength=6).astype(np.int64)
    need = [t for t in range(6) if masstab[t] == 0 and tot[t] > 0]
    if need and mass is None:
        raise ValueError("per-particle masses needed for types %s" % need)
    if tot[0] > 0 and u is None:
        raise ValueError("internal energies needed for the gas particles")
    m_sorted = np.asarray(mass, dtype=np.float64)[order] if mass is not None else None
    u_sorted = np.asarray(u, dtype=np.float64)[order][: tot[0]] if u is not None else None
    start = np.concatenate([[0], np.cumsum(tot)])[:6]
    # share of every type in every file
    share = np.zeros((num_files, 6), dtype=np.int64)
    for t in range(6):
        cuts = (np.arange(num_files + 1) * tot[t]) // num_files
        share[:, t] = np.diff(cuts)
    offs = np.zeros(6, dtype=np.int64)
    paths = []
    for f_i in range(num_files):
        path = base if num_files == 1 else "%s.%d" % (base, f_i)
        paths.append(path)
        sel = np.concatenate([order[start[t] + offs[t]: start[t] + offs[t] + share[f_i, t]] for t in range(6)]) \
            if share[f_i].sum() else np.zeros(0, dtype=np.int64)
        t_sel = ptype[sel]
        blocks = [("POS", pos[sel].astype("<f4").tobytes()), ("VEL", vel[sel].astype("<f4").tobytes()),
                  ("ID", ids[sel].astype("<u4").tobytes())]
        if need:
            mm = np.asarray(mass, dtype=np.float64)[sel]
            blocks.append(("MASS", np.concatenate([mm[t_sel == t] for t in need]).astype("<f4").tobytes()))
        if share[f_i, 0] > 0:
            blocks.append(("U", u_sorted[offs[0]: offs[0] + share[f_i, 0]].astype("<f4").tobytes()))
        with open(path, "wb") as f:
            def rec(label, payload):
                if snap_format == 2:
                    f.write(struct.pack("<i", 8) + _LABELS[label].encode() + struct.pack("<i", len(payload) + 8) + struct.pack("<i", 8))
                f.write(struct.pack("<i", len(payload)) + payload + struct.pack("<i", len(payload)))
            rec("HEAD", _header_bytes(share[f_i], masstab, time, boxsize, tot, num_files))
            for label, payload in blocks:
                if len(payload):
                    rec(label, payload)
        offs += share[f_i]
    return paths


def _read_one(path):
    with open(path, "rb") as f:
        raw = f.read()
    off = 0
    fmt2 = struct.unpack_from("<i", raw, 0)[0] == 8        # a format-2 file starts with the 8-byte label record

    def block(expect=None):
        nonlocal off
        if fmt2:
            (nb,) = struct.unpack_from("<i", raw, off)
            label = raw[off + 4: off + 8].decode()
            if nb != 8 or struct.unpack_from("<i", raw, off + 12)[0] != 8:
                raise ValueError("corrupt format-2 label record at %d" % off)
            off += 16
            if expect and label != _LABELS[expect]:
                raise ValueError("expected block %r, found %r" % (_LABELS[expect], label))
        (nb,) = struct.unpack_from("<i", raw, off)
        payload = raw[off + 4: off + 4 + nb]
        if struct.unpack_from("<i", raw, off + 4 + nb)[0] != nb:
            raise ValueError("corrupt record at %d" % off)
        off += 8 + nb
        return payload

    h = block("HEAD")
    npart = np.frombuffer(h, dtype="<i4", count=6, offset=0).astype(np.int64)
    masstab = np.frombuffer(h, dtype="<f8", count=6, offset=24).copy()
    time, redshift = struct.unpack_from("<dd", h, 72)
    tot_lo = np.frombuffer(h, dtype="<u4", count=6, offset=96).astype(np.int64)
    num_files = struct.unpack_from("<i", h, 124)[0]
    boxsize = struct.unpack_from("<d", h, 128)[0]
    tot_hi = np.frombuffer(h, dtype="<u4", count=6, offset=176).astype(np.int64)
    n = int(npart.sum())
    pos = np.frombuffer(block("POS"), dtype="<f4", count=3 * n).reshape(n, 3).astype(np.float64)
    vel = np.frombuffer(block("VEL"), dtype="<f4", count=3 * n).reshape(n, 3).astype(np.float64)
    ids = np.frombuffer(block("ID"), dtype="<u4", count=n).copy()
    ptype = np.repeat(np.arange(6, dtype=np.int32), npart)
    mass = np.repeat(masstab, npart).astype(np.float64)
    need = [t for t in range(6) if masstab[t] == 0 and npart[t] > 0]
    if need:
        mblk = np.frombuffer(block("MASS"), dtype="<f4").astype(np.float64)
        k = 0
        s0 = np.concatenate([[0], np.cumsum(npart)])
        for t in need:
            mass[s0[t]: s0[t] + npart[t]] = mblk[k: k + npart[t]]
            k += npart[t]
    u = None
    if npart[0] > 0 and off < len(raw):
        u = np.frombuffer(block("U"), dtype="<f4", count=int(npart[0])).astype(np.float64)
    header = dict(npart=npart, mass=masstab, time=time, redshift=redshift, boxsize=boxsize, num_files=num_files,
                  npart_total=tot_lo + (tot_hi << 32), format=2 if fmt2 else 1)
    return dict(pos=pos, vel=vel, ids=ids, mass=mass, type=ptype, u=u, header=header)


def read_snapshot(base):
    """Reads a format-1 or format-2 snapshot / IC (detected from the first record), single file `base` or the set
    base.0 ... base.(num_files-1); particles come back grouped by type across the files, as read_ic.c places them."""
    first = base if os.path.exists(base) else base + ".0"
    d0 = _read_one(first)
    nf = max(1, d0["header"]["num_files"])
    parts = [d0] + [_read_one("%s.%d" % (base, i)) for i in range(1, nf)] if first != base else [d0]
    if len(parts) == 1:
        if d0["header"]["num_files"] <= 1:
            d0["header"]["npart_total"] = d0["header"]["npart"].copy()      # read_ic.c:323-326
        return d0
    out = {}
    for key in ("pos", "vel", "ids", "mass", "type"):
        out[key] = np.concatenate([np.concatenate([p[key][p["type"] == t] for p in parts]) for t in range(6)])
    us = [p["u"] for p in parts if p["u"] is not None]
    out["u"] = np.concatenate(us) if us else None
    hdr = dict(d0["header"])
    hdr["npart"] = sum(p["header"]["npart"] for p in parts)
    out["header"] = hdr
    return out
