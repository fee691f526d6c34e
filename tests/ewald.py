"""Independent truth for periodic boxes: Ewald summation of the Newtonian force plus nearest-image
sums of the (exponentially screened) Yukawa part.  Used to judge TreePM = tree + PM totals the way
the reference's FORCETEST does with its Ewald-corrected direct sum (forcetree.c:3428-3548,
gravtree_forcetest.c:28-356); written from the textbook formulas, no reference tables involved.
"""
import numpy as np
from scipy.special import erfc


def _plummer_fac(m, h, r):
    """spline factor (already /r), ngravs.c:420-434 form"""
    u = r / h
    lo = 10.666666666667 + u * u * (32.0 * u - 38.4)
    with np.errstate(divide="ignore", invalid="ignore"):
        hi = 21.333333333333 - 48.0 * u + 38.4 * u * u - 10.666666666667 * u ** 3 - 0.066666666667 / u ** 3
    return m / h ** 3 * np.where(u < 0.5, lo, hi)


def ewald_direct(pos, mass, species, idx, L, G, law, ym, h):
    """acceleration (xG) of targets idx.  law[tg][sg] in {0 none,1 newton,2 -newton,3 yukawa,4 coloyuk};
    ym = YUKAWA_IMASS/BoxSize; h = softening length (one value)."""
    alpha = 2.0 / L
    nimg = np.array([(i, j, k) for i in range(-2, 3) for j in range(-2, 3) for k in range(-2, 3)], dtype=float) * L
    hv = np.array([(i, j, k) for i in range(-4, 5) for j in range(-4, 5) for k in range(-4, 5)
                   if 0 < i * i + j * j + k * k <= 16], dtype=float)
    kv = 2 * np.pi / L * hv
    k2 = (kv ** 2).sum(1)
    kfac = 4 * np.pi / L ** 3 * np.exp(-k2 / (4 * alpha * alpha)) / k2
    near = np.array([(i, j, k) for i in range(-1, 2) for j in range(-1, 2) for k in range(-1, 2)], dtype=float) * L
    law = np.asarray(law)
    out = np.zeros((len(idx), 3))
    for t, i in enumerate(idx):
        d = pos - pos[i]
        d -= L * np.round(d / L)                       # nearest image
        tg = species[i]
        lw = law[tg][species]                          # law against every source
        cN = np.where((lw == 1) | (lw == 4), 1.0, np.where(lw == 2, -1.0, 0.0))
        cY = np.where((lw == 3) | (lw == 4), 1.0, 0.0)
        acc = np.zeros(3)
        # Newtonian part: real-space images
        for n in nimg:
            dd = d + n
            r = np.sqrt((dd ** 2).sum(1))
            with np.errstate(divide="ignore", invalid="ignore"):
                f = (erfc(alpha * r) + 2 * alpha * r / np.sqrt(np.pi) * np.exp(-(alpha * r) ** 2)) / r ** 3
            f[r == 0] = 0.0
            acc += ((cN * mass * f)[:, None] * dd).sum(0)
        # k-space
        ph = d @ kv.T                                   # N x K
        s = (cN * mass) @ np.sin(ph)                    # K
        acc += (kfac * s) @ kv
        # Yukawa part: nearest images only (exp(-ym L) ~ e^-60)
        if np.any(cY != 0):
            for n in near:
                dd = d + n
                r = np.sqrt((dd ** 2).sum(1))
                with np.errstate(divide="ignore", invalid="ignore"):
                    f = np.exp(-r * ym) * (ym / r + 1 / r ** 2) / r
                f[r == 0] = 0.0
                acc += ((cY * mass * f)[:, None] * dd).sum(0)
        # softened pairs (nearest image only): replace the point-mass law by the spline
        r0 = np.sqrt((d ** 2).sum(1))
        m = (r0 < h) & (r0 > 0)
        if np.any(m):
            rr = r0[m]
            point = cN[m] / rr ** 3 + cY[m] * np.exp(-rr * ym) * (ym / rr + 1 / rr ** 2) / rr
            spl = np.where(lw[m] != 0, 1.0, 0.0) * np.where(lw[m] == 2, -1.0, 1.0) * _plummer_fac(1.0, h, rr)
            acc += ((mass[m] * (spl - point))[:, None] * d[m]).sum(0)
        out[t] = acc * G
    return out
