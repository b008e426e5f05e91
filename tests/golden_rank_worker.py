"""Worker of tests/test_golden_hmc.py::test_gpu_replays_reference_trajectory_over_real_ranks: the reference's HMC regression run
(golden set G7, tests/extra/staghmc_sh/ref.*) with the 8^4 lattice split along t over the launched ranks -- processes that share
device 0, faces and rank sums through the peer-memory transport -- held to the reference's printed numbers at its own tolerance
by the same checker the single-rank replay uses (test_golden_hmc._check_trajectory).  Every rank checks; GOLDEN_RANK_OK per rank."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    run, resident = int(sys.argv[1]), bool(int(sys.argv[2]))
    import torch.distributed as dist

    dist.init_process_group("gloo")
    world, rank = dist.get_world_size(), dist.get_rank()
    import qex_amd as q
    from oracle import oracle as o
    import hmc_replay as R
    import test_golden_hmc as T

    ranks = (world, rank, dist)
    lt = R.LAT[3] // world
    be = R.HipBackend(q, R.LAT, resident=resident, ranks=ranks)
    assert be.ctx.comm_transport()[0] == "peer"
    rng = q.RngField(R.LAT[:3] + [lt], q.RngMilc6, R.SEED, glat=R.LAT, t_offset=rank * lt)      # newRNGField: seeded by global site index
    r = R.Replay(o, be, R.CONFIGS[run], rng=rng, ranks=ranks)
    T._check_trajectory(r, second=(run == 0 and not resident))
    st = be.ctx.comm_transport()[1]
    be.ctx.close()
    for k in range(world):
        if k == rank:
            sys.stdout.write("\nGOLDEN_RANK_OK rank %d run %d resident %d exchanges %d allreduces %d\n" % (rank, run, resident, st["exchanges"], st["allreduces"]))
            sys.stdout.flush()
        dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
