"""The fork's HMC program (src/stagg_pv_hmc/staghmc_spv.nim) as examples/staghmc_spv.py, every field operation in
libqexhip.  The reference tree holds no golden log for this program, so it is held to invariants: the run-time check
the reference itself performs (reversibility, staghmc_spv.nim:1091-1160), dH ~ dt^2 (which fails unless each force is
the gradient of the action it is paired with: gauge, smeared gauge, fermion and Pauli-Villars sectors all enter), and
agreement between the device-resident MD loop, the host-field entry points and the sharded kernels."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))

LAT = [8, 4, 4, 8]
PRM = dict(start="0.25", g_steps=6, sg_steps=3, f_steps=3, pv_steps=2, tau=0.5, seed=4711)


def _traj(resident=True, halo=False, **kw):
    import staghmc_spv as S

    p = dict(PRM)
    p.update(kw)
    h = S.Spv(LAT, resident=resident, halo=halo, **p)
    h.refresh()
    b = h.action()
    h.evolve()
    e = h.action()
    return h, b, e


@pytest.mark.gpu
def test_gpu_spv_trajectory_resident_host_and_sharded_agree():
    hr, br, er = _traj(resident=True)
    hh, bh, eh = _traj(resident=False)
    hs, bs, es = _traj(resident=True, halo=True)
    for k in ("H", "ga", "sga", "fa", "T"):
        assert abs(br[k] - bh[k]) <= 1e-12 * abs(br["H"]) and abs(br[k] - bs[k]) <= 1e-12 * abs(br["H"])
        assert abs(er[k] - eh[k]) <= 1e-10 * abs(er["H"]), (k, er[k], eh[k])
        assert abs(er[k] - es[k]) <= 1e-10 * abs(er["H"]), (k, er[k], es[k])
    assert np.abs(hr.g - hh.g).max() < 1e-11 and np.abs(hr.p - hh.p).max() < 1e-10
    assert np.abs(hr.g - hs.g).max() < 1e-11
    # every sector contributes to the start energy (nothing silently switched off)
    assert br["sga"] != 0.0 and len(br["f2"]) == 3 and all(v > 0 for v in br["f2"]) and br["ga"] != 0.0


@pytest.mark.gpu
def test_gpu_spv_trajectory_is_reversible_and_second_order():
    h, b, e = _traj()
    g_end = h.g.copy()
    dH1 = e["H"] - b["H"]
    # reversibility (rev_check, staghmc_spv.nim:1091-1160): flip the momenta, evolve again, back at the start
    h.p *= -1.0
    h.evolve()
    r = h.action()
    assert abs(r["H"] - b["H"]) <= 1e-9 * abs(b["H"])
    assert np.abs(h.g - g_end).max() > 1e-3           # it did move
    # the same trajectory with twice the steps in every sector: dH drops by ~4 (2MN is second order)
    _, b2, e2 = _traj(g_steps=12, sg_steps=6, f_steps=6, pv_steps=4)
    dH2 = e2["H"] - b2["H"]
    assert abs(b2["H"] - b["H"]) <= 1e-12 * abs(b["H"])                 # same start (same seed)
    assert abs(dH1) > 1e-6 and abs(dH2) < abs(dH1) / 2.8, (dH1, dH2)


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2])
def test_gpu_spv_trajectory_over_real_ranks(nranks):
    """BASELINE configs[4]'s program with REAL neighbours: the same trajectory (nHYP closure, smeared gauge force, fermion and
    Pauli-Villars forces with their solves, 2MN schedule, resident MD) with the lattice split along t over processes that share
    the one GPU (peer transport) -- per-site random streams seeded by global index, so the fields are those of the single-rank
    run -- must give the single-rank energies, sector by sector, and the same end links on every slab."""
    import json
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hr, br, er = _traj(resident=True)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", QEXHIP_PEER_TIMEOUT="60")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1",
           "--master-port", "29577", os.path.join(root, "tests", "spv_rank_worker.py")] + [str(v) for v in LAT] + [json.dumps(PRM)]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=root, env=env)
    rows = [json.loads(ln.split(" ", 1)[1]) for ln in p.stdout.splitlines() if ln.startswith("SPV_RANK ")]
    if p.returncode != 0 or len(rows) != nranks:
        print(p.stdout[-3000:])
        print(p.stderr[-6000:])
    assert p.returncode == 0 and len(rows) == nranks
    lt = LAT[3] // nranks
    for r in rows:
        for k in ("H", "ga", "sga", "fa", "T"):
            assert abs(r["begin"][k] - br[k]) <= 1e-12 * abs(br["H"]), (k, r["begin"][k], br[k])
            assert abs(r["end"][k] - er[k]) <= 1e-10 * abs(er["H"]), (k, r["end"][k], er[k])
        # this rank's slab of the end links: the single-rank field cut along t (V=1 even-odd order: select by coordinates)
        import qex_amd as q
        _, idx = q.Layout(LAT).shard_indices(nranks, r["rank"])
        slab = hr.g[idx]
        assert abs(r["g_sum"] - float((slab * slab).sum())) <= 1e-9 * abs(r["g_sum"])
        assert np.abs(np.array(r["g_first"]) - slab.reshape(-1)[:6]).max() < 1e-10
    assert rows[0]["begin"]["H"] == rows[1]["begin"]["H"]                 # every rank holds the same rank-summed numbers
