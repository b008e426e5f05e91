"""Worker of tests/test_gpu_two_ranks.py::test_peer_transport_failures_are_errors_not_hangs: two ranks share device 0 (peer transport).

  scenario "absent":   rank 1 joins the communicator and then leaves without ever exchanging; rank 0 sets links (collective: the
                       receive arena is sized in a host rendezvous) -> QEXHIP_ERR_COMM from the bounded host barrier.
  scenario "vanish":   both ranks set links and apply the operator once, then rank 1 leaves; rank 0 applies it again.  Every
                       device-side wait of the transport is bounded (QEXHIP_PEER_TIMEOUT): rank 0's exchange kernel gives up
                       waiting for the neighbour's faces, and the next host sync returns QEXHIP_ERR_COMM with a message that
                       names the wait, in about that time -- not a hang, not a fault.
  scenario "mismatch": rank 1 asks for QEXHIP_TRANSPORT=rccl (no rendezvous), rank 0 for auto: nobody else shows up in rank 0's
                       rendezvous, which since round 6 is what a job that spans nodes looks like -- after QEXHIP_RENDEZVOUS_TIMEOUT
                       rank 0 takes RCCL too (one decision for the job) and joins rank 1's ncclCommInitRank.  On this one-GPU box
                       RCCL then refuses the duplicate device: an error from RCCL on both ranks, in bounded time, not a hang and
                       not a rendezvous error.
Prints PEER_FAILURE_OK from rank 0 on the expected behaviour."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    scenario = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if scenario == "mismatch" and rank == 1:
        os.environ["QEXHIP_TRANSPORT"] = "rccl"
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    import qex_amd as q

    lat = [8, 8, 8, 4]
    ctx = q.Context(lat, device=0, rank_geom=(1, 1, 1, world), rank_coord=(0, 0, 0, rank))
    uid = [q.Context.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    if scenario == "mismatch":
        t0 = time.time()
        try:
            ctx.comm_init(uid[0], world, rank)
            what = "communicator on " + ctx.comm_transport()[0]        # (two GPUs: this is the outcome)
            assert ctx.comm_transport()[0] == "rccl", what
        except q.QexHipError as e:
            what = str(e)
            assert "rendezvous" not in what and ("nccl" in what.lower() or "rccl" in what.lower()), what
        dt = time.time() - t0
        assert dt < 90, dt
        if rank == 0:
            assert dt > 3.0, dt                   # it did wait for the rendezvous first
            print("PEER_FAILURE_OK mismatch after %.1f s: rank 0 followed rank 1 to RCCL: %s" % (dt, what[:160]), flush=True)
        return
    ctx.comm_init(uid[0], world, rank)
    assert ctx.comm_transport()[0] == "peer"
    rng = np.random.default_rng(5 + rank)
    vol = int(np.prod(lat))
    g = rng.standard_normal((vol, 4, 3, 3, 2))
    x = rng.standard_normal((vol, 3, 2))
    dist.barrier()
    if scenario == "absent":
        if rank == 1:
            return                              # gone: no set_links, no exchange, no close
        t0 = time.time()
        try:
            q.newStag(ctx, g)                   # collective: the backward t-links come from the lower rank -- which never sends
            print("rank 0: set_links returned although the neighbour never took part", flush=True)
            sys.exit(2)
        except q.QexHipError as e:
            dt = time.time() - t0
            assert "did not reach barrier" in str(e) and dt < 30, (str(e), dt)
            print("PEER_FAILURE_OK absent after %.1f s: %s" % (dt, str(e)[:160]), flush=True)
        return
    s = q.newStag(ctx, g)
    r = np.zeros_like(x)
    s.stagD2(r, x, "all", 0.0, 0.0)             # a complete exchange: arenas sized, both directions used
    ctx.sync()
    dist.barrier()
    if rank == 1:
        time.sleep(1.0)
        return                                  # gone, its context still open: nothing marks the segment failed
    t0 = time.time()
    try:
        s.stagD2(r, x, "all", 0.0, 0.0)
        ctx.sync()
        print("rank 0: the operator returned although the neighbour never exchanged", flush=True)
        sys.exit(2)
    except q.QexHipError as e:
        dt = time.time() - t0
        assert "timed out" in str(e) and 2.0 < dt < 30, (str(e), dt)
        print("PEER_FAILURE_OK vanish after %.1f s: %s" % (dt, str(e)[:200]), flush=True)


if __name__ == "__main__":
    main()
