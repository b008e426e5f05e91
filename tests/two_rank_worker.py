"""Worker of tests/test_gpu_two_ranks.py: one process per GPU, launched by torch.distributed.run BEFORE anything touches a
GPU, real RCCL between two (or more) distinct devices.

Every rank builds the same GLOBAL problem with the oracle (gauge field, sources), hands its t-slab to
libqexhip -- context with rankGeom {1,1,1,N}, communicator from qexhip_comm_init -- and checks ITS slab of every result
against the global oracle result:

  stagD2 on the three subsets / D           <= 1e-13   (faces through ncclSend/ncclRecv, shifts.nim:67-94,254-285, qshifts.nim:51-131)
  CG residual history (solveEE)             first 100 iterations 1e-10, iteration count +-1, solution 1e-6 (cg.nim:174-217:
                                            both reductions end in a rank sum, commsUtils.nim:195-204)
  lock-step batched CG (3 systems)          iteration counts +-1, solutions 1e-6 (all faces in one RCCL group)
  Naik 3-mass multi-shift CG                the same bars, ghost depth 3
  plaquettes, Polyakov loops, one flow step 1e-13 / 1e-13 / 1e-12 (ghost links refreshed per stage; the t-line through every rank's slab)
  nHYP smearing + smeared gauge force       1e-11

usage (by the test): python -m torch.distributed.run --nproc-per-node N two_rank_worker.py LX LY LZ LT [--overlap K]
Exit status 0 and a line `TWO_RANK_OK rank r ...` per rank, non-zero on the first failed check.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SEED = 987654321


def relerr(a, b):
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))


def main():
    import time
    t_start = time.time()

    def tick(what):
        if os.environ.get("QEX_WORKER_VERBOSE"):
            print("[%6.1f s] %s" % (time.time() - t_start, what), file=sys.stderr, flush=True)

    ap = argparse.ArgumentParser()
    ap.add_argument("lat", type=int, nargs=4)
    ap.add_argument("--overlap", type=int, default=-1, help="option overlap of the context: -1 by size, 0 never, 1 always")
    ap.add_argument("--hop-split", type=int, default=-1, help="option hop_split: 2 = the fused self-pushing sweep, 0 = split by sites, -1 = measured")
    ap.add_argument("--fused-spin-us", type=int, default=-1, help="option fused_spin_us: -2 = every boundary block parked")
    ap.add_argument("--skip-gauge", action="store_true", help="operator and solvers only")
    ap.add_argument("--share-device", action="store_true", help="every rank binds device 0: the peer-memory transport between processes "
                    "that share one GPU (RCCL refuses that), i.e. real neighbours on a one-GPU box")
    args = ap.parse_args()
    rank, world, local_rank = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("NCCL_DEBUG", "WARN")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)      # control plane only (unique id, final barrier)
    import qex_amd as q
    from oracle import oracle as o

    glat = list(args.lat)
    olo = o.Layout(glat)
    rf = o.RngField(olo, o.RNG_MILC6, SEED)
    g0 = o.gauge_random(olo, rf)                                      # unphased: the gauge sector's input
    g = g0.copy()
    o.rephase(olo, g)
    g3 = o.gauge_random(olo, rf)
    o.rephase(olo, g3)
    g3 *= 0.3
    x, y = o.vector_gaussian(olo, rf), o.vector_gaussian(olo, rf)
    tick("imports + oracle inputs done")
    loc, idx = q.Layout(glat).shard_indices(world, rank)
    vh = loc.vol // 2
    dev = 0 if args.share_device else local_rank

    if world > 1:
        ctx = q.Context(loc.lat, device=dev, rank_geom=(1, 1, 1, world), rank_coord=(0, 0, 0, rank))
    else:
        ctx = q.Context(loc.lat, device=dev)                    # one-GPU rehearsal of this script: the sharded code path ...
    uid = [q.Context.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    ctx.comm_init(uid[0], world, rank)
    if world == 1:
        ctx.force_halo(True)                                           # ... ghost zones filled through a one-rank communicator,
        ctx.set_option("multi_reduce", 1)                              # reductions through real (one-rank) all-reduces
        ctx.set_option("batch_multi", 1)                               # ... in the lock-step batched CG too
    tick("context + comm_init done")
    info = ctx.comm_info()
    assert info[0] == world and info[1] == rank, info                  # RCCL's own count and rank
    transport = ctx.comm_transport()[0]
    wish = os.environ.get("QEXHIP_TRANSPORT", "auto")
    if (args.share_device and world > 1) or wish == "peer":
        assert transport == "peer", transport
    elif wish in ("mbox", "rccl"):
        assert transport == {"mbox": "rccl+mbox", "rccl": "rccl"}[wish], transport
    else:
        # auto between distinct devices: RCCL for the faces; the mailboxes for the rank sums if they passed comm_init's self-test
        assert transport in ("rccl+mbox", "rccl"), transport
    if args.overlap >= 0:
        ctx.set_option("overlap", args.overlap)
    if args.hop_split >= 0:
        ctx.set_option("hop_split", args.hop_split)
    if args.fused_spin_us != -1:
        ctx.set_option("fused_spin_us", args.fused_spin_us)
    res = {"rank": rank, "device": info[2], "pci_bus": info[3], "comms": ctx.comm_count(), "transport": transport}

    def sl(a):
        return np.ascontiguousarray(a[idx])

    # ---- one-hop operator ----
    s = q.newStag(ctx, sl(g))
    worst = 0.0
    for subset, par in (("even", 0), ("odd", 1), ("all", 2)):
        for a, b in ((0.0, 0.0), (0.3, 0.7)):
            ref = y.copy()
            o.stagD2(olo, g, None, ref, x, par, a, b)
            r = sl(y)
            s.stagD2(r, sl(x), subset, a, b)
            worst = max(worst, relerr(r, sl(ref)))
    r = np.zeros_like(sl(x))
    s.D(r, sl(x), 0.1)
    worst = max(worst, relerr(r, sl(o.D(olo, g, None, x, 0.1))))
    res["stagD2_D"] = worst
    res["sweep"] = ctx.sweep_info()                                    # incl. the measured overlap decision (world > 1, option -1)
    if os.environ.get("QEX_WORKER_VERBOSE"):
        print("rank %d sweep %s" % (rank, json.dumps(res["sweep"])), file=sys.stderr, flush=True)
    assert worst < 1e-13, worst

    tick("operator done")
    # ---- CG (solveEE): history, count, solution ----
    sp = q.SolverParams(r2req=1e-12, maxits=5000, verbosity=0)
    xs = np.zeros_like(sl(x))
    s.solveEE(xs, sl(x), 0.1, sp, histcap=8192)
    xr, its, fin, hist = o.solveXX(olo, g, None, x, 0.1, 1e-12, 5000, True, histcap=8192)
    n = min(len(hist), len(sp.r2hist))
    dev = np.abs(sp.r2hist[:n] / hist[:n] - 1)
    res["cg"] = {"its": sp.iterations, "oracle_its": int(its), "dev100": float(dev[:100].max()), "dev_all": float(dev.max()),
                 "x": relerr(xs[:vh], sl(xr)[:vh])}
    assert abs(sp.iterations - its) <= 1 and res["cg"]["dev100"] < 1e-10 and res["cg"]["x"] < 1e-6, res["cg"]

    # ---- lock-step batched CG (the HMC's Hasenbusch solves): all systems' faces in one RCCL group, n scalars per all-reduce ----
    bms = [0.1, 0.2, 0.4]
    bbs = [sl(x), sl(y), sl(x) + sl(y)]
    for b in bbs:
        b[vh:] = 0
    bxs = [np.zeros_like(b) for b in bbs]
    bits, _ = s.solveXX_batch(bxs, bbs, bms, 1e-12, 5000, True)
    worst = 0.0
    for k, (src, m) in enumerate(zip((x, y, x + y), bms)):
        xr_k, its_k, _, _ = o.solveXX(olo, g, None, src, m, 1e-12, 5000, True)
        assert abs(bits[k] - its_k) <= 1, (k, list(bits), its_k, res.get("sweep"))
        worst = max(worst, relerr(bxs[k][:vh], sl(xr_k)[:vh]))
    res["batch"] = {"its": list(bits), "x": worst}
    assert worst < 1e-6, res["batch"]

    tick("CG + batched CG done")
    # ---- Naik: stagD2 and the 3-mass multi-shift CG (ghost depth 3) ----
    if loc.lat[3] >= 4:
        s3 = q.newStag3(ctx, sl(g), sl(g3))
        ref = y.copy()
        o.stagD2(olo, g, g3, ref, x, 2, 0.3, 0.7)
        r = sl(y)
        s3.stagD2(r, sl(x), "all", 0.3, 0.7)
        res["naik_stagD2"] = relerr(r, sl(ref))
        assert res["naik_stagD2"] < 1e-13, res["naik_stagD2"]
        masses = [0.1, 0.2, 0.4]
        shifts = [masses[0]] + [4.0 * (m * m - masses[0] ** 2) for m in masses[1:]]      # stagSolve.nim:391-394
        ys = [np.zeros_like(sl(x)) for _ in masses]
        spm = q.SolverParams(r2req=1e-12, maxits=5000, verbosity=0)
        s3.solveXX_multi(ys, sl(x), shifts, spm, histcap=8192)
        yr, mits, mhist = o.solveXX_multi(olo, g, g3, x, shifts, 1e-12, 5000, True, histcap=8192)
        n = min(len(mhist), len(spm.r2hist))
        dev = np.abs(spm.r2hist[:n] / mhist[:n] - 1)
        res["naik_multishift"] = {"its": spm.iterations, "oracle_its": int(mits), "dev100": float(dev[:100].max()),
                                  "x": max(relerr(a[:vh], sl(b)[:vh]) for a, b in zip(ys, yr))}
        assert abs(spm.iterations - mits) <= 1 and res["naik_multishift"]["dev100"] < 1e-10 and res["naik_multishift"]["x"] < 1e-6, res["naik_multishift"]

    # ---- gauge sector: plaquettes, one flow step, nHYP smear + smeared gauge force ----
    if not args.skip_gauge:
        pl = q.plaq(ctx, sl(g0))
        res["plaq"] = float(np.abs(pl - o.plaq(olo, g0)).max())
        assert res["plaq"] < 1e-13, res["plaq"]
        # the four Polyakov loops: the t-line runs through EVERY rank's slab (one segment product per rank, rank-ordered
        # all-gather, gauge.hip k_tline_segment / k_tline_trace), the spatial lines stay inside a slab and are rank-summed
        pls = q.ploops(ctx)
        ref_pl = [o.wline(olo, g0, [d + 1] * glat[d]) for d in range(4)]
        res["ploops"] = float(max(abs(a - b) for a, b in zip(pls, ref_pl)))
        assert res["ploops"] < 1e-13, (pls, ref_pl)
        gf = sl(g0)
        q.gaugeFlow(ctx, gf, 1, 0.01)
        gr = g0.copy()
        o.wflow(olo, gr, 1, 0.01)
        res["wflow_links"] = relerr(gf, sl(gr))
        res["wflow_plaq"] = float(np.abs(q.plaq(ctx) - o.plaq(olo, gr)).max())
        assert res["wflow_links"] < 1e-12 and res["wflow_plaq"] < 1e-13, res
        gw = o.gauge_warm(olo, 0.5, rf)
        sg = np.zeros_like(sl(gw))
        sf = q.HypCoefs(0.4, 0.5, 0.5).smearGetForce(ctx, sl(gw), sg)
        sgr = o.nhyp_smear(olo, gw, 0.4, 0.5, 0.5)
        sgr = sgr[0] if isinstance(sgr, tuple) else sgr
        res["nhyp_smear"] = relerr(sg, sl(sgr))
        f = np.zeros_like(sg)
        sf.gforce(f, plaq=1.0)
        chain = o.gauge_deriv_general(olo, sgr, 1.0, 0.0, 0)
        _, fr = o.nhyp_force(olo, gw, chain, 0.4, 0.5, 0.5)
        o.force_projTAH(olo, fr, gw, adj=True)
        res["nhyp_gforce"] = relerr(f, sl(fr))
        # the operator on the closure's links, BC + staggered phases put on by the DEVICE (k_rephase: the t boundary condition
        # lives on the last rank only -- the one piece of rank-coordinate logic a one-rank run cannot reach)
        sn = q.Staggered(ctx, None, smear=q.HypCoefs(0.4, 0.5, 0.5), bc="pppa")
        sgp = sgr.copy()
        o.rephase(olo, sgp)
        r = np.zeros_like(sl(x))
        sn.D(r, sl(x), 0.1)
        res["D_on_device_rephased_nhyp_links"] = relerr(r, sl(o.D(olo, sgp, None, x, 0.1)))
        assert res["D_on_device_rephased_nhyp_links"] < 1e-12, res
        sf.release()
        assert res["nhyp_smear"] < 1e-12 and res["nhyp_gforce"] < 1e-11, res

    tick("Naik + gauge sector done")
    res["transport_stats"] = ctx.comm_transport()[1]
    ctx.close()
    dist.barrier()
    for r in range(world):                      # one rank at a time: lines of concurrent writers run into each other
        if r == rank:
            sys.stdout.write("\nTWO_RANK_OK rank %d %s\n" % (rank, json.dumps(res)))
            sys.stdout.flush()
        dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
