"""Round 6, VERDICT item 1(a): where does the fused self-pushing sweep (hop_split = 2) first fail when the ranks SHARE a GPU?

One process per rank, all on device 0, peer-memory transport, overlap forced, a K-iteration CG (2K sweeps) on random
links -- no oracle, so big slabs cost nothing on the CPU.  The same solve is run with hop_split = 0 (split by sites) first:
its residual history is the reference the fused run must reproduce (to rounding: the hop split sums local hops first).

  python tests/shared_device_worker.py LX LY LZ LT_GLOBAL [--naik] [--its K]     (RANK / WORLD_SIZE / MASTER_* from the env)

Prints one line `BISECT rank r {...}` with ok / the library's error text / the number of boundary workgroups per sweep.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lat", type=int, nargs=4)
    ap.add_argument("--naik", action="store_true")
    ap.add_argument("--its", type=int, default=40)
    ap.add_argument("--forms", type=int, nargs="+", default=[0, 2], help="hop_split values to run, in order")
    ap.add_argument("--distinct-devices", action="store_true", help="rank r binds device LOCAL_RANK instead of device 0 (scratch/first_contact.sh)")
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    import qex_amd as q

    loc = list(args.lat)
    loc[3] //= world
    vol = int(np.prod(loc))
    rng = np.random.default_rng(1234 + rank)
    dev = int(os.environ.get("LOCAL_RANK", rank)) % max(q.device_count(), 1) if args.distinct_devices else 0
    ctx = q.Context(loc, device=dev, rank_geom=(1, 1, 1, world), rank_coord=(0, 0, 0, rank))
    uid = [q.Context.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    ctx.comm_init(uid[0], world, rank)
    ctx.set_option("overlap", 1)
    g = (0.35 * rng.standard_normal((vol, 4, 3, 3, 2))).astype(np.float64)
    g3 = (0.1 * rng.standard_normal((vol, 4, 3, 3, 2))).astype(np.float64) if args.naik else None
    b = rng.standard_normal((vol, 3, 2))
    b[vol // 2:] = 0
    F = loc[0] * loc[1] * loc[2] // 2
    depth = 3 if args.naik else 1
    res = {"rank": rank, "local": loc, "naik": bool(args.naik), "boundary_workgroups": 2 * depth * ((F + 255) // 256),
           "sweep_workgroups": (vol // 2 + 255) // 256, "transport": None, "forms": {}}
    hist0 = None
    for form in args.forms:
        ctx.set_option("hop_split", form)
        t0 = time.time()
        try:
            s = q.newStag3(ctx, g, g3) if args.naik else q.newStag(ctx, g)
            res["transport"] = ctx.comm_transport()[0]
            sp = q.SolverParams(r2req=0.0, maxits=args.its, verbosity=0)
            x = np.zeros_like(b)
            s.solveEE(x, b, 0.5, sp, histcap=args.its + 8)
            ctx.sync()
            h = np.array(sp.r2hist[: args.its + 1])
            out = {"ok": True, "its": int(sp.iterations), "s": round(time.time() - t0, 2)}
            if hist0 is None:
                hist0 = h
            else:
                n = min(len(h), len(hist0))
                out["hist_dev_vs_first_form"] = float(np.abs(h[:n] / hist0[:n] - 1).max())
        except Exception as e:  # noqa: BLE001
            out = {"ok": False, "error": str(e)[:400], "s": round(time.time() - t0, 2)}
        res["forms"]["hop_split=%d" % form] = out
        print("BISECT-PROGRESS rank %d %s form %d %s" % (rank, "x".join(map(str, args.lat)), form, json.dumps(out)), file=sys.stderr, flush=True)
        if not out["ok"]:
            break
    res["stats"] = ctx.comm_transport()[1]
    sys.stdout.write("\nBISECT rank %d %s\n" % (rank, json.dumps(res)))
    sys.stdout.flush()
    # no collective teardown: after a timeout the peer may be gone
    os._exit(0 if all(v["ok"] for v in res["forms"].values()) else 3)


if __name__ == "__main__":
    main()
