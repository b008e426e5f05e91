"""Host-side gauge-field helpers that stay on the host in QEX as well: boundary conditions,
staggered phases, and synthetic configurations for benchmarks.

    setBC      src/gauge/gaugeUtils.nim:124-131   U_3 *= -1 on the last t slice
    stagPhase  src/physics/stagD.nim:509-520      eta_mu from bit masks [8,9,11,0]
    rephase    src/physics/stagD.nim:72-80        setBC then stagPhase

These are sign flips on host arrays (numpy); the arithmetic of the hot path is in libqexhip.
"""
import numpy as np


def unit(lo):
    g = lo.newGauge()
    for i in range(3):
        g[:, :, i, i, 0] = 1.0
    return g


def setBC(lo, g, t_offset=0, t_global=None):
    """Anti-periodic t boundary.  For a t-slab of a sharded lattice pass the slab's global
    t offset and the global extent."""
    T = t_global if t_global is not None else lo.lat[3]
    last = (lo.coords[:, 3] + t_offset) == T - 1
    g[last, 3] *= -1.0


def stagPhase(lo, g, phases=(8, 9, 11, 0), t_offset=0):
    x = lo.coords.copy()
    x[:, 3] += t_offset
    for mu in range(4):
        s = np.zeros(lo.vol, dtype=np.int64)
        for k in range(4):
            s += (phases[mu] >> k) & x[:, k]
        g[(s & 1) == 1, mu] *= -1.0


def rephase(lo, g, t_offset=0, t_global=None):
    setBC(lo, g, t_offset, t_global)
    stagPhase(lo, g, t_offset=t_offset)


def synthetic_random_su3(lo, seed=987654321, spread=None, chunk=1 << 18):
    """Synthetic SU(3) configuration for benchmarks (numpy; NOT QEX's RngMilc6 stream -- the
    oracle reproduces that one for the parity tests).  Rows 0,1 of a complex Gaussian matrix are
    orthonormalised (Gram-Schmidt), row 2 = conj(row0 x row1), which gives det = 1 exactly.
    With `spread`, links are close to unity (a 'warm' start, better conditioned).  Generated in
    chunks to bound the temporary memory (a 48^3x96 field is 6 GB)."""
    rng = np.random.default_rng(seed)
    n = lo.vol * 4
    g = np.empty((n, 3, 3, 2))
    for i0 in range(0, n, chunk):
        m = min(chunk, n - i0)
        a = rng.standard_normal((m, 2, 3)) + 1j * rng.standard_normal((m, 2, 3))
        if spread is not None:
            a = np.eye(3)[None, :2] + spread * a
        r0 = a[:, 0] / np.linalg.norm(a[:, 0], axis=1)[:, None]
        r1 = a[:, 1] - np.sum(r0.conj() * a[:, 1], axis=1)[:, None] * r0
        r1 /= np.linalg.norm(r1, axis=1)[:, None]
        r2 = np.conj(np.cross(r0, r1))
        q = np.stack([r0, r1, r2], axis=1)
        g[i0:i0 + m, :, :, 0] = q.real
        g[i0:i0 + m, :, :, 1] = q.imag
    return g.reshape(lo.vol, 4, 3, 3, 2)


def synthetic_gaussian_vector(lo, seed=12345):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((lo.vol, 3, 2))


def repeat_in_t(lob, field, n):
    """A field on the block lattice `lob` (even-odd site order, t slowest within each parity half, even t extent)
    repeated n times along t: the same field on the lattice with n times the t extent, in that lattice's site order."""
    if lob.lat[3] % 2:
        raise ValueError("block t extent must be even (the site parity must repeat)")
    vh = lob.vol // 2
    reps = (n,) + (1,) * (field.ndim - 1)
    return np.concatenate([np.tile(field[:vh], reps), np.tile(field[vh:], reps)])


def synthetic_repeated_su3(lat, t_block, seed=987654321, t_offset=0, t_global=None):
    """Benchmark configuration for large lattices: one random SU(3) block of `t_block` slices repeated along t
    (numpy needs about a minute for 10 M independent sites), already rephased (antiperiodic boundary on the last
    global slice + staggered phases, which have period 2 in t).  Returns the field of the local lattice `lat`."""
    from .layout import Layout

    if lat[3] % t_block or t_block % 2 or t_offset % 2:
        raise ValueError("t extents must be even multiples of the block")
    lob = Layout(list(lat[:3]) + [t_block])
    gb = synthetic_random_su3(lob, seed=seed)
    stagPhase(lob, gb)
    g = repeat_in_t(lob, gb, lat[3] // t_block)
    T = t_global if t_global is not None else lat[3]
    if t_offset + lat[3] == T:                       # this slab holds the last global slice
        vh = g.shape[0] // 2
        f = lat[0] * lat[1] * lat[2] // 2
        for par in range(2):
            g[par * vh + vh - f:(par + 1) * vh, 3] *= -1.0
    return g
