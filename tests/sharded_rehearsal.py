"""Worker of tests/test_host_logic.py::test_sharded_dslash_two_ranks_gloo.

CPU rehearsal of the t-sharded Dslash (one process per rank, gloo): the slab decomposition, the
ghost-zone layout (positions from the library's own index code, qexhip_debug_nbr_pos), the
one-time ghost-link exchange and the per-sweep face exchange in the message order of
qex_amd/csrc/comm.cpp -- with numpy doing the 3x3 arithmetic -- must reproduce the oracle's
global stagD2 on every slab.  usage: sharded_rehearsal.py RANK WORLD NAIK
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, naik = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    import qex_amd as q
    from oracle import oracle as o

    L = q.lib()
    glat = [8, 8, 8, 4 * world]
    olo = o.Layout(glat)
    rf = o.RngField(olo, o.RNG_MILC6, 987654321)
    g = o.gauge_random(olo, rf)
    o.rephase(olo, g)
    g3 = None
    if naik:
        g3 = o.gauge_random(olo, rf)
        o.rephase(olo, g3)
        g3 *= 0.3
    x = o.vector_gaussian(olo, rf)
    y = o.vector_gaussian(olo, rf)
    a, b = 0.3, 0.7
    ref = y.copy()
    o.stagD2(olo, g, g3, ref, x, 2, a, b)

    cx = lambda arr: arr[..., 0] + 1j * arr[..., 1]
    loc, idx = q.Layout(glat).shard_indices(world, rank)
    lat = loc.lat
    i4 = (C.c_int * 4)(*lat)
    depth = 3 if naik else 1
    out = (C.c_int * 8)()
    assert L.qexhip_debug_geom(i4, depth, 1, out) == 0
    vh, F = out[0], out[1]
    n = depth * F
    up, down = (rank + 1) % world, (rank - 1 + world) % world

    xl, yl = cx(x[idx]), cx(y[idx])
    links = [cx(g[idx])] + ([cx(g3[idx])] if naik else [])

    def exchange(send_lo, send_hi, recv_hi, recv_lo):
        """comm.cpp::comm_halo_exchange order: send bottom->lower, top->upper; recv hi<-upper, lo<-lower."""
        ts = [torch.from_numpy(np.ascontiguousarray(v)) for v in (send_lo, send_hi)]
        tr = [torch.zeros_like(ts[0]), torch.zeros_like(ts[1])]
        ops = [dist.P2POp(dist.isend, ts[0], down), dist.P2POp(dist.isend, ts[1], up),
               dist.P2POp(dist.irecv, tr[0], up), dist.P2POp(dist.irecv, tr[1], down)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        recv_hi[...] = tr[0].numpy()
        recv_lo[...] = tr[1].numpy()

    # fields with ghost zones: [parity][pos][colour]
    fe = np.zeros((2, vh + 2 * n, 3), dtype=complex)
    for p in (0, 1):
        fe[p, :vh] = xl[p * vh:(p + 1) * vh]
        bottom, top = fe[p, :n].copy(), fe[p, vh - n:vh].copy()
        rh, rl = np.zeros((2, n, 3)), np.zeros((2, n, 3))
        exchange(np.stack([bottom.real, bottom.imag]), np.stack([top.real, top.imag]), rh, rl)
        fe[p, vh:vh + n] = rh[0] + 1j * rh[1]
        fe[p, vh + n:vh + 2 * n] = rl[0] + 1j * rl[1]

    # ghost links (layout.hip::links_upload): top `depth` slices of U_3, both parities, sent up
    ghosts = []
    for lk in links:
        send = np.zeros((2, n, 3, 3), dtype=complex)
        for p in (0, 1):
            send[p] = lk[p * vh + vh - n:p * vh + vh, 3]
        t_s = torch.from_numpy(np.stack([send.real, send.imag]))
        t_r = torch.zeros_like(t_s)
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, t_s, up), dist.P2POp(dist.irecv, t_r, down)]):
            w.wait()
        ghosts.append(t_r[0].numpy() + 1j * t_r[1].numpy())

    res = np.zeros((2 * vh, 3), dtype=complex)
    for p in (0, 1):
        for c in range(vh):
            i = p * vh + c
            acc = a * yl[i] + b * xl[i]
            t = c // F
            for li, lk in enumerate(links):
                h = 3 if li else 1
                for mu in range(4):
                    pf = L.qexhip_debug_nbr_pos(i4, depth, 1, c, p, mu, h)
                    acc = acc + lk[i, mu] @ fe[1 - p, pf]
                    pb = L.qexhip_debug_nbr_pos(i4, depth, 1, c, p, mu, -h)
                    if mu == 3 and t - h < 0:
                        U = ghosts[li][1 - p, (t - h + depth) * F + (c - t * F)]
                    else:
                        cb = L.qexhip_debug_nbr_pos(i4, depth, 0, c, p, mu, -h)
                        U = lk[(1 - p) * vh + cb, mu]
                    acc = acc - U.conj().T @ fe[1 - p, pb]
            res[i] = acc
    want = cx(ref[idx])
    err = np.linalg.norm(res - want) / np.linalg.norm(want)
    dist.barrier()
    dist.destroy_process_group()
    assert err < 1e-13, err
    print(f"SHARDED_OK rank {rank} err {err:.2e}")


if __name__ == "__main__":
    main()
