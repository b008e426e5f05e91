"""Real RCCL between DISTINCT GPUs: skipped on a one-GPU box, decisive on anything larger.

The ranks are started by torch.distributed.run as fresh processes (nothing is re-exec'ed, this pytest process never hands
its GPU state to them); every rank checks its t-slab of each result against the GLOBAL oracle result
(tests/two_rank_worker.py).  Replaces shifts.nim:67-94,254-285 / qshifts.nim:51-131 (paired send/recv of the faces) and
commsUtils.nim:195-204 (rank sums) -- the one part of the product a single-GPU box cannot exercise.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ngpu():
    import torch

    return torch.cuda.device_count()          # counts without initialising the GPU on this image


def _free_port():
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _launch(nranks, script_args, port=None, timeout=600, extra_env=None):
    port = port or _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS=str(max(1, min(16, len(os.sched_getaffinity(0))) // nranks)))   # a GPU box leases 16 cores whatever cpu_count() says
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, cwd=ROOT, env=env)


needs2 = pytest.mark.skipif("_ngpu() < 2", reason="needs two GPUs: real RCCL transport between distinct devices")


# ---- real neighbours on ONE device: the peer-memory transport (csrc/peer.hip) between processes that share GPU 0 ----
@pytest.mark.parametrize("nranks,lat,overlap,form", [(2, [8, 8, 8, 8], -1, -1), (2, [8, 8, 8, 8], 1, -1), (2, [16, 16, 16, 32], -1, -1),
                                                     (2, [16, 16, 16, 32], 1, 2), (2, [16, 16, 16, 32], 1, 0), (2, [16, 16, 16, 32], 1, -2),
                                                     (4, [8, 8, 8, 16], -1, -1), (4, [8, 8, 8, 16], 1, -1), (4, [16, 16, 16, 32], 1, 2),
                                                     (4, [16, 16, 16, 32], 1, 0), (4, [16, 16, 16, 32], 1, -2)])
def test_ranks_sharing_one_device_against_the_global_oracle(nranks, lat, overlap, form):
    """Every rank is its own process with its own slab, neighbours are OTHER processes: rank > 0 kernels, backward t-links
    fetched from the lower rank, the last-rank-only boundary condition of k_rephase, collective set_links, chunk agreement --
    everything a one-rank rehearsal cannot reach -- checked slab by slab against the global oracle (tests/two_rank_worker.py).
    Faces and reductions go through hipIpc-mapped peer memory (qshifts.nim:51-131, shifts.nim:67-94,254-285,
    commsUtils.nim:195-204 are what that replaces).  form: -1 the library's own choice (the fused self-pushing sweep unless set_links
    measures the split by sites faster), 2 fused, 0 split by sites, -2 fused with every boundary block PARKED (the cleanup workgroups
    do all slab-leaving hops: the path a late neighbour puts a block on)."""
    extra = [] if form == -1 else (["--hop-split", "2", "--fused-spin-us", "-2"] if form == -2 else ["--hop-split", str(form)])
    p = _launch(nranks, [os.path.join(ROOT, "tests", "two_rank_worker.py")] + [str(v) for v in lat] +
                ["--overlap", str(overlap), "--share-device"] + extra, extra_env={"QEXHIP_PEER_TIMEOUT": "60"})
    ok = [ln for ln in p.stdout.splitlines() if ln.startswith("TWO_RANK_OK")]
    if p.returncode != 0 or len(ok) != nranks:
        print(p.stdout[-4000:])
        print(p.stderr[-8000:])
    assert p.returncode == 0 and len(ok) == nranks, (p.returncode, len(ok))
    res = [json.loads(ln.split(" ", 3)[3]) for ln in ok]
    assert {r["transport"] for r in res} == {"peer"} and len({r["pci_bus"] for r in res}) == 1
    assert all(r["transport_stats"]["exchanges"] > 100 and r["transport_stats"]["allreduces"] > 100 for r in res), res[0]["transport_stats"]
    if form in (0, 2, -2) and overlap == 1:
        assert all(r["sweep"]["form"] == ("by_sites" if form == 0 else "fused") for r in res), res[0]["sweep"]
    assert len({json.dumps(r["sweep"]["tuned_us_per_sweep"]) for r in res}) == 1        # the measurement is ONE decision for the job


def test_fused_sweep_with_more_boundary_workgroups_than_the_chip_has_slots():
    """The failure round 5 left open (profiles/r06_notes.md section 1): two ranks SHARING the chip, fused sweeps, a face with more
    boundary workgroups than the chip has slots for the kernel -- 16 links on 40^3 x 8 slabs: 750 of them against 768 slots, 48^3: 1296.
    Boundary workgroups that SPIN until the faces are in then keep the neighbour's push (another process's kernel) from ever becoming
    resident: 0x510 after QEXHIP_PEER_TIMEOUT in round 5's build.  Since round 6 they wait about one exchange time, park, and leave; the
    solve must run through and reproduce the split-by-sites history to rounding (tests/shared_device_worker.py, no oracle at this size)."""
    for lat, naik in (([40, 40, 40, 16], True), ([48, 48, 48, 24], True), ([64, 64, 64, 16], False)):
        p = _launch(2, [os.path.join(ROOT, "tests", "shared_device_worker.py")] + [str(v) for v in lat] + (["--naik"] if naik else []) + ["--its", "24"],
                    timeout=200, extra_env={"QEXHIP_PEER_TIMEOUT": "20"})
        ok = [json.loads(ln.split(" ", 3)[3]) for ln in p.stdout.splitlines() if ln.startswith("BISECT rank")]
        if p.returncode != 0 or len(ok) != 2:
            print(p.stdout[-3000:])
            print(p.stderr[-6000:])
        assert p.returncode == 0 and len(ok) == 2, (lat, p.returncode)
        for r in ok:
            f = r["forms"]["hop_split=2"]
            assert f["ok"] and f["hist_dev_vs_first_form"] < 1e-12, (lat, r)


@pytest.mark.parametrize("scenario", ["absent", "vanish", "mismatch"])
def test_peer_transport_failures_are_errors_not_hangs(scenario):
    """A neighbour that never exchanges, or ranks that disagree about the transport, must end in QEXHIP_ERR_COMM with a message,
    within the configured bounds -- every device-side wait of the peer transport has an exit (tests/peer_failure_worker.py)."""
    p = _launch(2, [os.path.join(ROOT, "tests", "peer_failure_worker.py"), scenario], timeout=240,
                extra_env={"QEXHIP_PEER_TIMEOUT": "3", "QEXHIP_RENDEZVOUS_TIMEOUT": "4"})
    ok = [ln for ln in p.stdout.splitlines() if ln.startswith("PEER_FAILURE_OK")]
    if len(ok) != 1:
        print(p.stdout[-3000:])
        print(p.stderr[-6000:])
    assert len(ok) == 1, (p.returncode, len(ok))


@needs2
@pytest.mark.parametrize("lat,overlap", [([8, 8, 8, 8], -1), ([8, 8, 8, 8], 1), ([16, 16, 16, 32], -1), ([16, 16, 16, 32], 1)])
def test_two_ranks_against_the_global_oracle(lat, overlap):
    """8^4 (slabs of 4 slices: Naik ghost depth 3 of 4) and 16^3 x 32; overlap = 1 forces the interior / boundary split with
    the exchange on the second stream and its own communicator, -1 is the size-based default (one launch at these sizes)."""
    p = _launch(2, [os.path.join(ROOT, "tests", "two_rank_worker.py")] + [str(v) for v in lat] + ["--overlap", str(overlap)])
    ok = [ln for ln in p.stdout.splitlines() if ln.startswith("TWO_RANK_OK")]
    assert p.returncode == 0 and len(ok) == 2, (p.returncode, p.stdout[-1500:], p.stderr[-3000:])
    devs = {json.loads(ln.split(" ", 3)[3])["pci_bus"] for ln in ok}
    assert len(devs) == 2, devs                                    # two distinct devices


@needs2
@pytest.mark.parametrize("transport", ["rccl", "mbox", "peer"])
def test_two_ranks_on_each_transport(transport):
    """Two distinct devices, every transport by name (auto picks `mbox` there if its self-test passes): RCCL alone, RCCL faces + mailbox
    rank sums, and the peer-memory transport with its fused self-pushing sweep -- each rank's slab of every result against the global
    oracle (the same rungs scratch/first_contact.sh walks)."""
    p = _launch(2, [os.path.join(ROOT, "tests", "two_rank_worker.py"), "16", "16", "16", "32", "--overlap", "1"],
                extra_env={"QEXHIP_TRANSPORT": transport, "QEXHIP_PEER_TIMEOUT": "60"})
    ok = [ln for ln in p.stdout.splitlines() if ln.startswith("TWO_RANK_OK")]
    assert p.returncode == 0 and len(ok) == 2, (p.returncode, p.stdout[-1500:], p.stderr[-3000:])
    res = [json.loads(ln.split(" ", 3)[3]) for ln in ok]
    assert {r["transport"] for r in res} == {{"rccl": "rccl", "mbox": "rccl+mbox", "peer": "peer"}[transport]}
    assert len({r["pci_bus"] for r in res}) == 2


@needs2
def test_four_ranks_against_the_global_oracle():
    if _ngpu() < 4:
        pytest.skip("needs four GPUs")
    p = _launch(4, [os.path.join(ROOT, "tests", "two_rank_worker.py"), "8", "8", "8", "16", "--skip-gauge"])
    assert p.returncode == 0 and sum(ln.startswith("TWO_RANK_OK") for ln in p.stdout.splitlines()) == 4, (p.stdout[-1500:], p.stderr[-3000:])


@needs2
def test_bench_two_ranks_real_rccl():
    """the driver's N = 2 launch line: self-verification green through real transport, and the line explains itself"""
    p = _launch(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "5", "--repeats", "2"], timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    ln = json.loads([x for x in p.stdout.splitlines() if x.startswith("{")][-1])
    assert "error" not in ln and ln["rccl_nranks"] == 2 and ln["shard_check"]["ok"] is True, ln.get("shard_check")
    assert len({r["pci_bus"] for r in ln["ranks"]}) == 2
    d = ln["multi_gpu"]
    for k in ("interior_us", "boundary_us", "exchange_us", "allreduce_us", "comm_count", "overlap", "per_rank"):
        assert k in d, k
    leg = ln["cg_48x48x48x96"]
    assert leg["shard_check"]["ok"] is True and "multi_gpu" in leg


@pytest.mark.parametrize("lat,overlap", [([8, 8, 8, 8], 1), ([16, 16, 16, 32], -1)])
def test_worker_script_one_rank_rehearsal(lat, overlap):
    """The worker itself, run with ONE rank on the one GPU a test box has (sharded code path through a one-rank communicator):
    so that the first time two GPUs are available, what can fail is the transport and not the script."""
    p = _launch(1, [os.path.join(ROOT, "tests", "two_rank_worker.py")] + [str(v) for v in lat] + ["--overlap", str(overlap)])
    assert p.returncode == 0 and sum(ln.startswith("TWO_RANK_OK") for ln in p.stdout.splitlines()) == 1, (p.stdout[-1500:], p.stderr[-3000:])


def test_two_rank_tests_skip_cleanly_on_one_gpu():
    """this file must never fail for lack of hardware: on one GPU the RCCL launches above are skipped, not attempted (the
    condition is a string: pytest evaluates it at set-up, so collecting the file counts no devices)"""
    assert _ngpu() >= 1
    assert needs2.args[0] == "_ngpu() < 2"
