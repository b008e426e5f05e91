#!/usr/bin/env python
"""bench.py -- staggered Dslash + CG on a 32^4 SU(3) fp64 lattice (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W

One "step" = one iteration of the even/odd CG hot loop (src/solvers/cg.nim:174-214):
    p = r + beta p ; Ap = 4(m^2 - D_eo D_oe) p (two one-parity Dslash sweeps) ; <p,Ap> ;
    x += alpha p ; r -= alpha Ap ; |r|^2
on gauge links and vectors already resident in HBM.  The timed region is exactly K iterations
(r2req = 0 so the solver cannot stop early) between barrier + device synchronisation on both
sides; the time is the max over ranks.  `value` is the CG throughput in GFLOP/s with QEX's own
flop count, (4*nd*72+60) = 1212 flop per even site per iteration (src/physics/stagSolve.nim:92),
over the whole job; CG iterations/s and the Dslash GFLOP/s (570 flop/site) are reported next to
it, and the dominant kernel (the one-parity Dslash sweep) is priced against the HBM roofline from
hipEvent timings taken inside the timed region on the library's own stream.

N > 1: the 32^4 lattice is split along t over the N GPUs (strong scaling), faces exchanged with
RCCL send/recv overlapped with the interior sweep.  Launch with torch.distributed.run.

After the main measurement the same N GPUs run a short leg on BASELINE configs[3] (48^3 x 96, the lattice the
north star quotes its >= 6x strong-scaling target on); it is reported as "cg_48x48x48x96" inside the same JSON
line and can never cost the main line (exceptions are caught, a watchdog prints the line if the leg stalls).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
B_SWEEP1 = 1248                # bytes/site, stagDP sweep: 8 links*144 + read 48 + write 48
B_SWEEP2 = 1296                # stagDM sweep: + 48 for the 4m^2 x term (which also feeds <p,Ap>)
FLOP_DSLASH = 570              # 8*66 + 7*6 per site; QEX's own convention is 582
FLOP_CG = 1212                 # per even site per iteration, stagSolve.nim:92


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--lat", type=int, nargs=4, default=[32, 32, 32, 32])
    ap.add_argument("--mass", type=float, default=0.1)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--halo", action="store_true",
                    help="N=1 only: route t-hops through ghost zones + a one-rank RCCL communicator "
                         "(rehearses the sharded code path and its host overhead on one GPU)")
    ap.add_argument("--naik", action="store_true", help="add synthetic 3-hop (Naik) links: 16 links per site")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-48x96", action="store_true",
                    help="skip the extra leg on BASELINE configs[3] (48^3x96 CG over the same N GPUs)")
    ap.add_argument("--rehearse-no-rccl", action="store_true",
                    help="N>1 control-flow rehearsal on a box with fewer GPUs than ranks: every rank uses GPU 0 and wraps "
                         "its own slab periodically instead of talking to its neighbours (RCCL refuses duplicate GPUs); "
                         "the numbers are meaningless, only the launch / barrier / reporting path is exercised")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    N = args.gpus
    if world != N:
        if world == 1 and N > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {N} but WORLD_SIZE={world}")

    # multi-process GPU work on this pool needs dmabuf IPC (without it RCCL fails with hipIpcGetMemHandle: invalid
    # argument); RCCL problems should leave a trace in the log
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if N > 1:
        os.environ.setdefault("NCCL_DEBUG", "WARN")

    import torch
    import torch.distributed as dist

    def device_sync():
        # All GPU work of this benchmark runs on libqexhip's own HIP streams (system ROCm runtime,
        # bound with RTLD_DEEPBIND); PyTorch's bundled HIP runtime is a different instance and sees
        # none of it, so the authoritative synchronisation is ctx.sync().  torch.cuda.synchronize()
        # is added only if PyTorch's runtime happens to be initialised.
        ctx.sync()
        if torch.cuda.is_initialized():
            torch.cuda.synchronize()
    if N > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)  # control plane only

    import qex_amd as q

    lat = list(args.lat)
    if lat[3] % N or (lat[3] // N) % 2:
        raise SystemExit("t extent must split into even slabs")
    lt = lat[3] // N
    lat_loc = lat[:3] + [lt]
    lo = q.Layout(lat_loc)
    V = int(np.prod(lat))
    Vh_loc = lo.vol // 2

    # synthetic inputs (no network, no stored configurations): random SU(3) links with antiperiodic
    # t boundary + staggered phases, gaussian source.
    g = q.synthetic_random_su3(lo, seed=987654321 + rank)
    q.rephase(lo, g, t_offset=rank * lt, t_global=lat[3])
    b = q.synthetic_gaussian_vector(lo, seed=4321 + rank)

    if args.rehearse_no_rccl:
        ctx = q.Context(lat_loc, device=0)
        ctx.force_halo(True)
    else:
        ctx = q.Context(lat_loc, device=local_rank, rank_geom=(1, 1, 1, N), rank_coord=(0, 0, 0, rank))
    if N > 1:
        uid = [q.Context.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        if not args.rehearse_no_rccl:
            ctx.comm_init(uid[0], N, rank)
    elif args.halo:
        ctx.comm_init(q.Context.unique_id(), 1, 0)
        ctx.force_halo(True)
    if args.naik:
        g3 = 0.3 * q.synthetic_random_su3(lo, seed=555 + rank)
        q.rephase(lo, g3, t_offset=rank * lt, t_global=lat[3])
        s = q.newStag3(ctx, g, g3)
    else:
        s = q.newStag(ctx, g)
    _, compressed, _ = s.links_info()           # SU(3)-up-to-sign links are streamed as 2 rows + sign bit
    nd = 8 if args.naik else 4                  # s.g.len in QEX's flop formulas
    b1 = 2 * nd * 144 + 96                      # bytes/site of a sweep: links once, vector in + out
    flop_dslash = 2 * nd * 66 + (2 * nd - 1) * 6
    flop_cg = 4 * nd * 72 + 60                  # stagSolve.nim:92
    bid = ctx.field_new(b)
    xid = ctx.field_new()

    def barrier():
        if N > 1:
            dist.barrier()

    def timed_cg(steps, warmup):
        ctx.dev_solve_xx(xid, bid, args.mass, 0.0, max(warmup, 1), True)      # warmup
        device_sync()
        ctx.timers_enable(int(os.environ.get("QEX_BENCH_TIMERS", "2")))  # 2: Dslash sweeps only
        ctx.timers_reset()
        barrier()
        device_sync()
        t0 = time.perf_counter()
        its, _, _ = ctx.dev_solve_xx(xid, bid, args.mass, 0.0, steps, True)
        device_sync()
        barrier()
        t1 = time.perf_counter()
        ctx.timers_enable(False)
        assert its == steps, (its, steps)
        return t1 - t0

    dt = timed_cg(args.steps, args.warmup)
    if N > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])

    n_int, ms_int = ctx.timer("dslash")
    n_bnd, ms_bnd = ctx.timer("dslash_bnd")
    n_blas, ms_blas = ctx.timer("blas")
    n_red, ms_red = ctx.timer("reduce")
    # dominant kernel: the one-parity Dslash sweep.  Per launch it processes the sites of that launch
    # (all Vh_loc without sharding; the interior range when the faces are split off).
    if N == 1 and not args.halo:
        sites_per_launch = Vh_loc
    else:
        # the library splits a sweep into interior + faces only when it overlaps the exchange with the
        # interior (interior >= 131072 sites, dslash.hip); otherwise one launch covers the slab
        F = lat_loc[0] // 2 * lat_loc[1] * lat_loc[2]
        interior = max(Vh_loc - 2 * (3 if args.naik else 1) * F, 0)
        ov = int(os.environ.get("QEXHIP_OVERLAP", "-1"))
        sites_per_launch = interior if (ov == 1 or (ov < 0 and interior >= 131072)) else Vh_loc
    avg_ms = ms_int / max(n_int, 1)
    b_alg = 0.5 * (b1 + b1 + 48) * sites_per_launch                     # 144 B per link
    b_streamed = b_alg - 2 * nd * {0: 0, 1: 48, 2: 32}[compressed] * sites_per_launch   # 96 / 112 B per compressed link
    achieved_gbs = b_alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    streamed_gbs = b_streamed / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # whole sweep incl. boundary launches, for the Dslash GFLOP/s figure
    sweep_ms = (ms_int + ms_bnd) / max(n_int, 1)
    dslash_gflops = flop_dslash * Vh_loc * N / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0

    out = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = flop_cg * (V // 2) * args.steps / dt / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "dslash_traffic.json")
        if os.path.exists(tf) and lat == [32, 32, 32, 32] and N == 1 and not args.naik and not args.halo:
            try:
                traffic = json.load(open(tf)).get({0: "hbm_bytes_per_launch_32x4", 1: "hbm_bytes_per_launch_32x4_recon12"}.get(compressed, "none"))
            except Exception:
                traffic = None
        out = {
            "metric": "staggered Dslash GFLOP/s + CG iters/s, 32^4 SU(3) fp64, 1/2/4/8 MI355X",
            "value": round(value, 2),
            "unit": "GFLOP/s",
            "n_gpus": N,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%dx%dx%dx%d SU(3) even/odd staggered CG (2 Dslash sweeps + BLAS per step), mass %g, "
                            "t split over %d GPU(s)" % (lat[0], lat[1], lat[2], lat[3], args.mass, N),
                "lattice": lat, "mass": args.mass, "parallelism": "t-shard x%d" % N,
                "links": {0: "18 reals (144 B/link)", 1: "su3 rows 0,1 + sign bit (96 B/link, row 2 rebuilt in registers)",
                          2: "u3 rows 0,1 + determinant (112 B/link, row 2 rebuilt in registers)"}[compressed],
            },
            "cg_iters_per_s": round(args.steps / dt, 2),
            "dslash_gflops": round(dslash_gflops, 1),
            "dslash_gflops_qex582": round(dslash_gflops * (6 + 2 * nd * 72) / flop_dslash, 1),
            "dslash_us_per_sweep": round(sweep_ms * 1e3, 2),
            "kernel_ms": {"dslash": round(ms_int, 3), "dslash_bnd": round(ms_bnd, 3), "blas": round(ms_blas, 3),
                          "reduce": round(ms_red, 3), "wall": round(dt * 1e3, 3)},
            "roofline": {
                "bound": "hbm", "kernel": "k_dslash (one-parity sweep)",
                "achieved": round(achieved_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved_gbs / HBM_PEAK_GBS, 4),
                "launches": n_int, "avg_us": round(avg_ms * 1e3, 2),
                "alg_bytes_per_launch": int(b_alg), "traffic": traffic,
                # what the kernel is asked to stream (= alg_bytes unless the links are compressed)
                "streamed_bytes_per_launch": int(b_streamed), "achieved_streamed": round(streamed_gbs, 1),
                "frac_streamed": round(streamed_gbs / HBM_PEAK_GBS, 4),
            },
        }
        if N == 1 and not args.halo:
            # three systems (the Hasenbusch chain 0.1 / 0.2 / 0.4 of tests/extra/staghmc_sh) in lock-step on the same
            # links: the links are streamed once per sweep for all of them (csrc/batch.hip); host fields, so the
            # figure is taken from the kernel timers of the iterations, not from the wall clock (PCIe in and out)
            ms3 = [0.1, 0.2, 0.4]
            bs3 = [b] + [q.synthetic_gaussian_vector(lo, seed=977 + k) for k in range(2)]
            xs3 = [np.zeros_like(v) for v in bs3]
            kb = max(args.steps // 4, 10)
            s.solveXX_batch(xs3, bs3, ms3, 0.0, 5, True)
            ctx.timers_enable(1)
            ctx.timers_reset()
            s.solveXX_batch(xs3, bs3, ms3, 0.0, kb, True)
            nd3, msd3 = ctx.timer("dslash_batch")
            _, msb3 = ctx.timer("blas")
            _, msr3 = ctx.timer("reduce")
            ctx.timers_enable(0)
            per = (msd3 + msb3 + msr3) / kb / 3.0           # ms per system-iteration (kernel time)
            out["batched_cg_3_systems"] = {
                "us_per_system_iteration": round(per * 1e3, 2), "cg_iters_per_s_per_system_kernel_time": round(1e3 / per / 1.0, 1),
                "sweep_us": round(1e3 * msd3 / max(nd3, 1), 2),
                "note": "kernel time of the lock-step iteration / 3; compare ms_per_step",
            }
        if N == 1 and compressed and not args.halo:
            # the same workload with link compression switched off (all 18 reals streamed)
            ctx.set_option("recon", 0)
            s = q.newStag3(ctx, g, g3) if args.naik else q.newStag(ctx, g)
            k2 = max(args.steps // 2, 10)
            dt2 = timed_cg(k2, args.warmup)
            n2, ms2 = ctx.timer("dslash")
            out["full18_links"] = {
                "cg_iters_per_s": round(k2 / dt2, 2), "value": round(flop_cg * (V // 2) * k2 / dt2 / 1e9, 2),
                "dslash_us_per_sweep": round(1e3 * ms2 / max(n2, 1), 2),
                "roofline_frac": round(b_alg / (ms2 / max(n2, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "note": "same links, compression off; also what the library picks by itself for QEX's g.random hot start "
                        "(projectSU of gaussians, unitary only to 1e-11) and for non-unitary (HISQ fat) links",
            }
            ctx.set_option("recon", 2)
            s = q.newStag3(ctx, g, g3) if args.naik else q.newStag(ctx, g)     # back to the default format
        if N == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(lat_loc, g, g3 if args.naik else None, b, args.mass, args.cpu_seconds)
    barrier()
    ctx.close()
    if lat == [32, 32, 32, 32] and not (args.no_48x96 or args.halo or args.naik):
        # BASELINE configs[3]: the lattice the north star quotes its strong-scaling target on.  Never allowed to cost
        # the main line: a watchdog prints it and ends the process if this leg stalls (e.g. one rank failing inside
        # a collective), and any exception only drops the leg.
        import threading

        done = threading.Event()

        def watchdog():
            if not done.wait(240.0):
                if rank == 0:
                    out["cg_48x48x48x96"] = {"error": "leg did not finish within 240 s"}
                    print(json.dumps(out), flush=True)
                os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            leg = leg_48x96(q, dist, torch, N, rank, local_rank, args)
        except Exception as e:          # noqa: BLE001 -- reported in the line, never fatal
            leg = {"error": repr(e)[:200]}
        done.set()
        if rank == 0:
            out["cg_48x48x48x96"] = leg
    if rank == 0:
        print(json.dumps(out), flush=True)
    if N > 1:
        dist.destroy_process_group()


def leg_48x96(q, dist, torch, N, rank, local_rank, args):
    """CG on 48^3 x 96 (BASELINE configs[3]) split along t over the same N GPUs: same step, same timing protocol
    (warmup, barrier + device sync on both sides, max over ranks) as the main measurement."""
    lat = [48, 48, 48, 96]
    lt = lat[3] // N
    lat_loc = lat[:3] + [lt]
    lo = q.Layout(lat_loc)
    # synthetic links: one random 48^3 x 12 block repeated along t (qex_amd/gauge.py: generating 10.6 M independent
    # sites with numpy would take a minute), rephased for this slab
    from qex_amd.gauge import synthetic_repeated_su3
    g = synthetic_repeated_su3(lat_loc, 12, seed=24680, t_offset=rank * lt, t_global=lat[3])
    b = q.synthetic_gaussian_vector(lo, seed=1357 + rank)
    if args.rehearse_no_rccl:
        ctx = q.Context(lat_loc, device=0)
        ctx.force_halo(True)
    else:
        ctx = q.Context(lat_loc, device=local_rank, rank_geom=(1, 1, 1, N), rank_coord=(0, 0, 0, rank))
    if N > 1:
        uid = [q.Context.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        if not args.rehearse_no_rccl:
            ctx.comm_init(uid[0], N, rank)
    s = q.newStag(ctx, g)
    del g
    bid = ctx.field_new(b)
    xid = ctx.field_new()
    steps = max(args.steps // 4, 20)
    ctx.dev_solve_xx(xid, bid, args.mass, 0.0, 5, True)
    ctx.sync()
    ctx.timers_enable(2)
    ctx.timers_reset()
    if N > 1:
        dist.barrier()
    ctx.sync()
    t0 = time.perf_counter()
    its, _, _ = ctx.dev_solve_xx(xid, bid, args.mass, 0.0, steps, True)
    ctx.sync()
    if N > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ctx.timers_enable(False)
    if N > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
    n_int, ms_int = ctx.timer("dslash")
    n_bnd, ms_bnd = ctx.timer("dslash_bnd")
    fmt = s.links_info()[1]
    ctx.close()
    V = int(np.prod(lat))
    return {
        "workload": "48x48x48x96 SU(3) even/odd staggered CG, mass %g, t split over %d GPU(s)" % (args.mass, N),
        "steps": int(its), "ms_per_step": round(dt / its * 1e3, 5), "cg_iters_per_s": round(its / dt, 2),
        "value": round(FLOP_CG * (V // 2) * its / dt / 1e9, 2), "unit": "GFLOP/s",
        "dslash_us_per_sweep_rank0": round(1e3 * (ms_int + ms_bnd) / max(n_int, 1), 2), "link_format": fmt,
        "note": "north-star scaling target: >= 6x at 8 GPUs relative to this leg at n_gpus = 1",
    }


def cpu_baseline(lat, g, g3, b, mass, budget_s):
    """The oracle's CG (plain C + OpenMP restatement of cg.nim / stagD.nim) on the same links and
    source, on the GPU box's host cores.  kind = "port": the reference itself is Nim and cannot be
    built here.  Bounded sample: calibrate on 2 iterations, then run ~budget_s worth."""
    from oracle import oracle as o

    lo = o.Layout(lat)
    t0 = time.perf_counter()
    o.solveXX(lo, g, g3, b, mass, 0.0, 2, True)
    per = (time.perf_counter() - t0) / 2.0
    n = int(max(3, min(400, budget_s / max(per, 1e-6))))
    t0 = time.perf_counter()
    _, its, _, _ = o.solveXX(lo, g, g3, b, mass, 0.0, n, True)
    dt = time.perf_counter() - t0
    vh = lo.vol // 2
    return {
        "value": round((4 * (8 if g3 is not None else 4) * 72 + 60) * vh * its / dt / 1e9, 3), "unit": "GFLOP/s",
        "cores": o.num_threads(),
        "kind": "port", "cg_iters_per_s": round(its / dt, 3),
        "sample": "%d CG iterations of the same %dx%dx%dx%d workload (same links/source), C+OpenMP oracle" % (
            its, lat[0], lat[1], lat[2], lat[3]),
    }


if __name__ == "__main__":
    main()
