"""Worker of tests/test_spv_hmc.py::test_gpu_spv_trajectory_over_real_ranks: one trajectory of examples/staghmc_spv.py with the
lattice split along t over the launched ranks (they share device 0: peer transport), energies printed by rank 0."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    import torch.distributed as dist

    dist.init_process_group("gloo")
    import staghmc_spv as S

    lat = [int(v) for v in sys.argv[1:5]]
    prm = json.loads(sys.argv[5])
    h = S.Spv(lat, resident=True, ranks=(dist.get_world_size(), dist.get_rank(), dist), **prm)
    assert h.ctx.comm_transport()[0] == "peer"
    h.refresh()
    b = h.action()
    h.evolve()
    e = h.action()
    # the end links of this rank's slab, as a checksum the single-rank run can be cut to
    out = {"rank": h.rank, "begin": {k: (v if k != "f2" else list(v)) for k, v in b.items()}, "end": {k: (v if k != "f2" else list(v)) for k, v in e.items()},
           "g_sum": float((h.g * h.g).sum()), "g_first": [float(v) for v in h.g.reshape(-1)[:6]], "iters": h.iters}
    for r in range(dist.get_world_size()):
        if r == h.rank:
            sys.stdout.write("\nSPV_RANK %s\n" % json.dumps(out))
            sys.stdout.flush()
        dist.barrier()
    h.ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
