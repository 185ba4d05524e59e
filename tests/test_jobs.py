"""The sharded job driver: job list -> LPT shards -> per-rank batches -> one gather, results in job
order; BASELINE configs 3 (forward screen: one receptor shared by every job) and 4 (target fishing: one ligand shared).

CPU part: batch planning, sharding, the shared-half assembly, and the whole driver over gloo with world_size 2 on a
stand-in sampler (host logic + gather only).  GPU part (-m gpu): HIP vs the CPU oracle on small cfg 3 / cfg 4 job lists
(scores and 20-step poses from the same initial poses and noise), and a 2-rank run on one GPU (gloo-staged gather) that
must reproduce the 1-rank result bit for bit in job order."""
import copy
import os
import socket
import subprocess
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from diffbindfr_amd import assemble, dist as ddist, synthetic

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def make_jobs(cfg_id, n_jobs, n_atoms, n_lig, seed=0):
    """Raw records + ComplexRecords of a small job list of the named BASELINE config: cfg 3 shares the PocketRecord,
    cfg 4 the LigandRecord, anything else shares nothing."""
    shared = synthetic.CONFIGS[cfg_id].get("shared")
    srng = np.random.default_rng(1000 * cfg_id + seed)
    sp = synthetic.make_pocket(srng, n_atoms) if shared == "receptor" else None
    sl = synthetic.make_ligand(srng, n_lig) if shared == "ligand" else None
    raw, halves = [], []
    for j in range(n_jobs):
        rng = np.random.default_rng([cfg_id, seed, j])
        pk = sp or synthetic.make_pocket(rng, int(round(n_atoms * rng.uniform(0.8, 1.2))))
        lg = sl or synthetic.make_ligand(rng, max(4, int(round(n_lig * rng.uniform(0.7, 1.3)))))
        raw.append(synthetic.make_record(pk, lg, np.random.default_rng([7, cfg_id, seed, 0 if shared else j])))
    pocket0 = assemble.PocketRecord(raw[0]) if shared == "receptor" else None
    lig0 = assemble.LigandRecord(raw[0]) if shared == "ligand" else None
    if shared == "receptor":          # one psi/chi record of the receptor for all jobs
        for r in raw[1:]:
            for k in ("torsion_angle",):
                r[k] = raw[0][k]
    jobs = [assemble.ComplexRecord(r, lig=lig0, pocket=pocket0) for r in raw]
    return raw, jobs


# ------------------------------------------------------------------------------------------------ CPU
def test_plan_batches_keeps_jobs_whole_and_balanced():
    b = ddist.plan_batches([(j, 40) for j in range(17)], 640)
    assert [sum(u[2] for u in x) for x in b] == [360, 320]
    assert sorted(u[0] for x in b for u in x) == list(range(17))
    b = ddist.plan_batches([(3, 1500), (5, 3)], 640)                 # an oversized job is cut at multiples of the batch size
    assert [u for x in b for u in x] == [(3, 0, 640), (3, 640, 640), (3, 1280, 220), (5, 0, 3)]
    assert all(sum(u[2] for u in x) <= 640 for x in b)
    assert ddist.plan_batches([], 640) == []


def test_shared_halves_are_assembled_once_and_equal_the_unshared_batch():
    for cfg in (3, 4):
        raw, jobs = make_jobs(cfg, 3, 60, 10)
        pb = assemble.assemble(jobs, [2, 1, 3], "cpu")
        ref = assemble.assemble([assemble.ComplexRecord(copy.deepcopy(r)) for r in raw], [2, 1, 3], "cpu")
        assert pb.dims == ref.dims
        for k in pb.t:
            assert torch.equal(pb.t[k], ref.t[k]), (cfg, k)
        shared = jobs[0].pocket if cfg == 3 else jobs[0].lig
        assert all((j.pocket if cfg == 3 else j.lig) is shared for j in jobs)
        assert len(shared._dev) == 1                                      # one cached upload per device


def test_shard_jobs_by_cost():
    raw, jobs = make_jobs(2, 7, 60, 10)
    shards, reps = ddist.shard_jobs(jobs, [4, 4, 8, 2, 4, 4, 6], 2)
    assert sorted(shards[0] + shards[1]) == list(range(7))
    load = [sum(jobs[j].cost * reps[j] for j in s) for s in shards]
    assert abs(load[0] - load[1]) <= max(j.cost * r for j, r in zip(jobs, reps))
    assert jobs[0].cost == ddist.complex_cost(jobs[0].n_a, jobs[0].n_l, jobs[0].n_cab)


class _StandInSampler:
    """Host-side stand-in with the driver-facing interface of DiffBindFRHIP (run_complexes, job_seed): poses are a
    deterministic function of the job's seed, so the gathered result can be checked without a GPU."""
    from diffbindfr_amd.sampler import DiffBindFRHIP as _D
    job_seed = staticmethod(_D.job_seed)

    def run_complexes(self, records, poses, device, tr_sigma_max, seeds, pose_ranges=None):
        lig, a14, lp, rp = [], [], [0], [0]
        for c, (r, n, s) in enumerate(zip(records, poses, seeds)):
            p0, pt = (0, n) if pose_ranges is None else pose_ranges[c]
            g = torch.Generator().manual_seed(s)          # the job's WHOLE tape, then the rows of this chunk (like draw_tapes)
            lig.append(torch.randn(pt, r.n_l, 3, generator=g)[p0:p0 + n].reshape(-1, 3))
            a14.append(torch.randn(pt, r.n_r, 14, 3, generator=g)[p0:p0 + n].reshape(-1, 14, 3))
            for _ in range(n):
                lp.append(lp[-1] + r.n_l)
                rp.append(rp[-1] + r.n_r)
        pb = SimpleNamespace(lig_ptr_host=torch.tensor(lp), res_ptr_host=torch.tensor(rp))
        return pb, torch.cat(lig)[None], torch.cat(a14)[None]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# the job tables of the multi-rank CPU tests: (config, jobs, poses per job, batch size)
_CASES = {
    "w2": (3, 5, [3, 2, 9, 1, 2], 4),                      # job 2 is larger than a batch
    "idle": (3, 5, [3, 2, 9, 1, 2], 4),                    # world 8, five jobs: three ranks have nothing to do
    "ragged17": (2, 17, [1 + (7 * j) % 5 for j in range(17)], 6),   # 17 jobs of unequal size and pose count
    "big": (4, 3, [2, 23, 1], 5),                          # one job of 23 poses in batches of 5
}


def _gloo_worker(rank, world, port, q, case, kw):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    ddist.init(backend="gloo")
    cfg, n_jobs, poses, bp = _CASES[case]
    raw, jobs = make_jobs(cfg, n_jobs, 60, 10)
    res = ddist.run_sharded(_StandInSampler(), jobs, poses, seed=11, device="cpu", batch_poses=bp, **kw)
    q.put((rank, [None if r is None else (r[0].numpy(), r[1].numpy()) for r in res]))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def _run_world(world, case, **kw):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, q, case, kw)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0            # nobody deadlocked, idle ranks included
    return [g[1] for g in got]


def _one_rank(case, batch_poses=None):
    cfg, n_jobs, poses, bp = _CASES[case]
    raw, jobs = make_jobs(cfg, n_jobs, 60, 10)
    return jobs, poses, ddist.run_sharded(_StandInSampler(), jobs, poses, seed=11, device="cpu", batch_poses=batch_poses or bp)


def _same(res, one):
    return all(np.array_equal(l, l1.numpy()) and np.array_equal(a, a1.numpy()) for (l, a), (l1, a1) in zip(res, one))


def test_run_sharded_over_gloo_world2_equals_one_rank():
    jobs, poses, one = _one_rank("w2")
    # job 2 has more poses than a batch holds: it is cut into chunks (0,4) (4,4) (8,1) that take their rows of the job's one tape
    assert [tuple(l.shape) for l, _ in one] == [(p, j.n_l, 3) for p, j in zip(poses, jobs)]
    assert not torch.equal(one[2][0][0], one[2][0][4])
    for res in _run_world(2, "w2"):         # every rank holds every job's poses, in job order, bit for bit
        assert _same(res, one)


def test_result_does_not_depend_on_the_batch_size():
    """a job cut into several batches used to seed every chunk separately, which made `batch_poses` part of the seed."""
    _, _, a = _one_rank("big", batch_poses=5)
    _, _, b = _one_rank("big", batch_poses=7)
    _, _, c = _one_rank("big", batch_poses=64)
    assert all(torch.equal(x[0], y[0]) and torch.equal(x[0], z[0]) and torch.equal(x[1], z[1]) for x, y, z in zip(a, b, c))


@pytest.mark.parametrize("case", ["idle", "ragged17", "big"])
def test_run_sharded_world8(case):
    """World 8 over gloo through the real run_sharded: five jobs (three idle ranks), 17 jobs of unequal pose counts, one job larger
    than a batch -- job order, bitwise equality with one rank, no deadlock of the ranks without work."""
    jobs, poses, one = _one_rank(case)
    shards, _ = ddist.shard_jobs(jobs, poses, 8)
    if case == "idle":
        assert sum(1 for s in shards if not s) == 3
    for res in _run_world(8, case):
        assert _same(res, one)


def test_gather_to_root_and_windows_and_host_store():
    """gather='root': rank 0 alone receives every job (dist.gather), the others keep their own; a window of 64 floats forces many
    collectives per gather; store='host' keeps the records in host memory.  All bitwise equal to the one-rank result."""
    jobs, poses, one = _one_rank("ragged17")
    shards, _ = ddist.shard_jobs(jobs, poses, 4)
    got = _run_world(4, "ragged17", gather="root", window_bytes=256, store="host")
    assert _same(got[0], one)
    for r in range(1, 4):
        mine = set(shards[r])
        for j, x in enumerate(got[r]):
            assert (x is not None) == (j in mine)
            if x is not None:
                assert np.array_equal(x[0], one[j][0].numpy()) and np.array_equal(x[1], one[j][1].numpy())
    for res in _run_world(3, "ragged17", gather="all", window_bytes=512):
        assert _same(res, one)


def test_record_halves_are_released_after_their_last_batch():
    cfg, n_jobs, poses, bp = _CASES["w2"]
    raw, jobs = make_jobs(cfg, n_jobs, 60, 10)          # cfg 3: one shared pocket
    seen = []

    class Spy(_StandInSampler):
        def run_complexes(self, records, poses, device, tr_sigma_max, seeds, pose_ranges=None):
            for r in records:
                r.lig.dev("cpu"), r.pocket.dev("cpu")
            seen.append(len(jobs[0].pocket.__dict__.get("_dev", {})))
            return super().run_complexes(records, poses, device, tr_sigma_max, seeds, pose_ranges)
    ddist.run_sharded(Spy(), jobs, poses, seed=1, device="cpu", batch_poses=bp)
    assert all(len(j.lig.__dict__.get("_dev", {})) == 1 for j in jobs)     # default: the caller's records are left as they were uploaded
    seen.clear()
    ddist.run_sharded(Spy(), jobs, poses, seed=1, device="cpu", batch_poses=bp, release=True)
    assert all(s == 1 for s in seen)                       # the shared pocket stays resident while batches still need it ...
    assert "_dev" not in jobs[0].pocket.__dict__           # ... and is dropped after the last one
    assert all("_dev" not in j.lig.__dict__ for j in jobs)


def test_release_auto_is_keyed_on_the_tables_bytes_against_free_device_memory(monkeypatch):
    """a job table whose records do not fit the device together must not stay resident by default."""
    cfg, n_jobs, poses, bp = _CASES["w2"]
    raw, jobs = make_jobs(cfg, n_jobs, 60, 10)
    total = jobs[0].pocket.nbytes() + sum(j.lig.nbytes() for j in jobs)        # the shared pocket counts once
    assert total > 0 and jobs[0].lig.nbytes() == sum(getattr(jobs[0].lig, k).numel() * getattr(jobs[0].lig, k).element_size() for k in jobs[0].lig._FIELDS)
    assert ddist.release_needed(jobs, "cpu") is False                           # the host: nothing to release
    monkeypatch.setattr(ddist, "free_device_bytes", lambda dev: 4 * total + 4)
    assert ddist.release_needed(jobs, "cpu") is False                           # a quarter of the free memory or less: stay resident
    monkeypatch.setattr(ddist, "free_device_bytes", lambda dev: 4 * total - 4)
    assert ddist.release_needed(jobs, "cpu") is True

    class Spy(_StandInSampler):
        def run_complexes(self, records, poses, device, tr_sigma_max, seeds, pose_ranges=None):
            for r in records:
                r.lig.dev("cpu"), r.pocket.dev("cpu")
            return super().run_complexes(records, poses, device, tr_sigma_max, seeds, pose_ranges)
    ddist.run_sharded(Spy(), jobs, poses, seed=1, device="cpu", batch_poses=bp)          # default "auto", table too large -> released
    assert "_dev" not in jobs[0].pocket.__dict__ and all("_dev" not in j.lig.__dict__ for j in jobs)


# ------------------------------------------------------------------------------------------------ GPU
def _hip(dev):
    import diffbindfr_amd as dba
    from oracle import score_model as sm
    mcfg = sm.default_cfg()
    params = sm.init_params(mcfg, seed=1)
    model = dba.TensorProductModelHIP({}).to(dev)
    model.load_state_dict(params, strict=True)
    return mcfg, params, model, dba.DiffBindFRHIP(diffusion_model=model, test_cfg={})


def _shared_config_vs_oracle(cfg_id):
    from oracle import sampler as osampler, schedule as osched, score_model as sm
    from tests.helpers import oracle_batch_from_packed, rel_err
    dev = torch.device("cuda:0")
    mcfg, params, model, samp = _hip(dev)
    raw, jobs = make_jobs(cfg_id, 3, 70, 12, seed=2)
    poses = [2, 1, 2]
    pb = assemble.assemble(jobs, poses, dev)
    init, z = samp.draw_tapes(jobs, poses, [samp.job_seed(5, j) for j in range(3)], dev, tr_sigma_max=3.0)
    assemble.init_poses(model, pb, init)
    torch.cuda.synchronize()
    d = oracle_batch_from_packed(raw, poses, pb)
    G = d.num_graphs
    # one score evaluation on the initial poses
    sc = osched.step_scalars(osched.default_sample_cfg(), 4)
    dd = osampler.set_time(copy.deepcopy(d), sc, G)
    ref = sm.forward(params, mcfg, copy.deepcopy(dd))
    scn, scm = dd.sc_tor_score_norm2, dd.sc_torsion_edge_mask.bool()
    out = model.score_packed(pb, dd.t, dd.tr_sigma, dd.rot_score_norm, dd.tor_score_norm2, scn[scm])
    for nm, a, b in zip(("tr", "rot", "tor", "sc_tor"), out, ref):
        assert rel_err(a, b) < 1e-4, (cfg_id, nm, rel_err(a, b))
    # the whole 20-step trajectory from the same poses and noise
    noise = SimpleNamespace(tr=z["tr"].cpu(), rot=z["rot"].cpu(), tor=z["tor"].cpu()[:, :int(d.tor_edge_mask.sum())],
                            sc=z["sc"].cpu()[:, :int(d.sc_torsion_edge_mask.sum())])
    T = synthetic.residue_tables()
    lig_ref, a14_ref = osampler.sample(params, mcfg, osched.default_sample_cfg(), copy.deepcopy(d), noise,
                                       torch.from_numpy(T["atom14_to_group"]).long())
    lig, a14 = samp.sample_packed(pb, z)
    assert (lig[0].cpu() - lig_ref[0]).norm(dim=-1).max() < 1e-3
    assert (a14[0].cpu() - a14_ref[0]).norm(dim=-1).max() < 1e-3
    # and through the job driver: same seeds => the same poses, in job order
    res = ddist.run_sharded(samp, jobs, poses, seed=5, device=dev, tr_sigma_max=3.0)
    lp = pb.lig_ptr_host.tolist()
    g = 0
    for j, p in enumerate(poses):
        assert torch.equal(res[j][0].reshape(-1, 3), lig[0, lp[g]:lp[g + p]]), j
        g += p


@pytest.mark.gpu
def test_cfg3_shared_receptor_vs_oracle():
    _shared_config_vs_oracle(3)


@pytest.mark.gpu
def test_cfg4_shared_ligand_vs_oracle():
    _shared_config_vs_oracle(4)


@pytest.mark.gpu
def test_unseeded_calls_differ_and_seeded_calls_repeat():
    dev = torch.device("cuda:0")
    _, _, _, samp = _hip(dev)
    _, jobs = make_jobs(2, 2, 50, 8)
    a = samp.sample_complexes(jobs, 2, device=dev)
    b = samp.sample_complexes(jobs, 2, device=dev)
    assert not torch.equal(a[0][0], b[0][0])
    c = samp.sample_complexes(jobs, 2, device=dev, seed=9)
    e = samp.sample_complexes(jobs, 2, device=dev, seed=9)
    assert all(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) for x, y in zip(c, e))
    # a job's poses do not depend on its batch mates
    f = samp.sample_complexes(jobs[1:], 2, device=dev, seed=9, job_ids=[1])
    assert torch.equal(f[0][0], c[2][0]) and torch.equal(f[1][1], c[3][1])


_RANK_SCRIPT = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from diffbindfr_amd import dist as ddist
from tests.test_jobs import make_jobs, _hip
rank, world, local = ddist.init()
dev = torch.device("cuda:0")
_, _, _, samp = _hip(dev)
raw, jobs = make_jobs(int(sys.argv[3]), 6, 70, 12, seed=4)
res = ddist.run_sharded(samp, jobs, [3, 2, 4, 1, 2, 3], seed=21, device=dev, batch_poses=6)
np.savez(sys.argv[2] + f".rank{rank}.npz", **{f"lig{j}": l.cpu().numpy() for j, (l, a) in enumerate(res)},
         **{f"a14_{j}": a.cpu().numpy() for j, (l, a) in enumerate(res)})
ddist.barrier()
if world > 1:
    import torch.distributed as dist
    dist.destroy_process_group()
'''


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_id", [3, 4])
def test_two_ranks_on_one_gpu_reproduce_one_rank_bit_for_bit(tmp_path, cfg_id):
    """torchrun-style spawn of 2 ranks sharing cuda:0 (records staged through the host for the gloo gather): every
    rank's gathered result must equal the 1-rank result, bitwise and in job order."""
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out1 = str(tmp_path / "one")
    subprocess.run([sys.executable, str(script), ROOT, out1, str(cfg_id)], check=True, timeout=900,
                   env=dict(base, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    out2 = str(tmp_path / "two")
    port = str(_free_port())
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, out2, str(cfg_id)],
                              env=dict(base, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r), MASTER_PORT=port,
                                       DBFR_DIST_BACKEND="gloo")) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    one = np.load(out1 + ".rank0.npz")
    for r in range(2):
        two = np.load(out2 + f".rank{r}.npz")
        assert sorted(two.files) == sorted(one.files)
        for k in one.files:
            assert np.array_equal(one[k], two[k]), (r, k)


_RCCL_SCRIPT = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from diffbindfr_amd import dist as ddist
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
local = torch.arange(1000, device=dev, dtype=torch.float32) * 0.5
for mode in ("all", "root"):                       # the driver's windowed gather exactly as run_sharded calls it under RCCL
    for window in (256, 4096):
        out = ddist._gather_windows(local, [1000], 1, 0, mode, window, dev)
        assert len(out) == 1 and torch.equal(out[0], local), (mode, window)
host = local.cpu().pin_memory()                    # store="host": pinned records staged through a device window
out = ddist._gather_windows(host, [1000], 1, 0, "all", 300, dev)
assert torch.equal(out[0], host)
buf = torch.empty(1000, device=dev)
dist.all_gather_into_tensor(buf, local)
assert torch.equal(buf, local)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)     # bench.py's max-over-ranks + barrier
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t) == 1.5
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl-ok")
'''


@pytest.mark.gpu
def test_rccl_backend_runs_the_drivers_collectives_on_one_rank(tmp_path):
    """The `nccl` (= RCCL) backend itself, as far as a 1-GPU box allows: a single-rank process group on cuda:0 through the very calls of
    the multi-GPU path -- `_gather_windows` in both modes (device records and pinned host records through a device window),
    `all_gather_into_tensor`, the float64 MAX all-reduce of bench.py's timing and the barrier.  (Two ranks cannot share one GPU under
    RCCL; the 2- and 8-rank runs of this file use gloo.)"""
    script = tmp_path / "rccl.py"
    script.write_text(_RCCL_SCRIPT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
def test_bench_line_through_rccl_on_one_rank():
    """`DBFR_DIST_SINGLE=1 python bench.py`: the whole N > 1 code path of bench.py -- process group, barriers, run_sharded's windowed
    all_gather of the real sampler's records, max over ranks -- with the `nccl` backend and ONE rank (what a 1-GPU box can run of it)."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0", DBFR_DIST_SINGLE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--batch-poses", "80",
                        "--no-cpu-baseline", "--no-latency", "--no-native", "--no-pmc"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(lines[0])
    assert line["config"]["dist_backend"] == "nccl" and line["n_gpus"] == 1 and line["value"] > 0
    assert line["config"]["poses_total"] == 80 and "backend=nccl" in r.stderr


@pytest.mark.gpu
def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it (the driver's own command form): the script starts its two ranks itself
    (LOCAL_RANK -> device, rendezvous on 127.0.0.1), rank 0's ONE line comes back through the parent, and the line shows that the
    collectives saw two ranks.  gloo, because two RCCL ranks cannot share the one GPU of this box."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(DBFR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch-poses", "320",
                        "--no-profile"], env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-4000:]
    out = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(out) == 1 and out[0].startswith('{"metric"'), r.stdout[-2000:]          # ONE line on stdout, nothing else
    line = json.loads(out[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    c = line["config"]
    assert c["dist_backend"] == "gloo" and c["ranks_seen"] == 2 and c["poses_total"] == 640
    assert sorted(p["rank"] for p in c["per_rank"]) == [0, 1] and sum(p["poses"] for p in c["per_rank"]) == 640
    assert all(p["elapsed_s"] > 0 for p in c["per_rank"])
    assert all(p["poses_per_sec_sampling"] > 0 and p["poses_per_sec_calibration_batch"] > 0 for p in c["per_rank"])
    assert 0.3 < c["parallel_efficiency"] <= 1.0 + 1e-6
    # the job table was sharded by the ranks' measured speed (one untimed calibration batch, all-gathered): weights around 1, within +-10 %
    assert len(c["rank_speed"]) == 2 and all(0.9 <= v <= 1.1 for v in c["rank_speed"]) and abs(sum(c["rank_speed"]) - 2.0) < 0.05
    # a launcher environment that contradicts the command line is refused, not silently run
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--no-profile"],
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_bench_launcher_relays_one_line_and_the_worst_exit_code(tmp_path, monkeypatch):
    """bench.spawn_ranks without a GPU: the children are this interpreter running a stand-in script; rank 0's result line is relayed
    alone, other output goes to stderr, a failing rank makes the launch fail."""
    import bench
    stand_in = tmp_path / "child.py"
    stand_in.write_text("import os, sys\n"
                        "r = int(os.environ['RANK']); assert os.environ['WORLD_SIZE'] == '3' and os.environ['LOCAL_RANK'] == str(r)\n"
                        "assert os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
                        "print('noise from rank', r)\n"
                        "if r == 0: print('{\"metric\": \"poses_per_sec\", \"n_gpus\": 3}')\n"
                        "sys.exit(int(os.environ.get('FAIL_RANK', '-1')) == r)\n")
    monkeypatch.setattr(bench, "__file__", str(stand_in))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    import contextlib, io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = bench.spawn_ranks(3, [])
    assert rc == 0 and buf.getvalue().strip() == '{"metric": "poses_per_sec", "n_gpus": 3}'
    monkeypatch.setenv("FAIL_RANK", "2")
    with contextlib.redirect_stdout(io.StringIO()):
        assert bench.spawn_ranks(3, []) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_id,n_jobs,poses,small_batch", [(2, 16, 40, 160), (5, 2, 20, 20)])
def test_full_size_batch_equals_small_batches_bit_for_bit(cfg_id, n_jobs, poses, small_batch):
    """BASELINE-sized batches (cfg 2: 16 complexes x 40 poses = the bench batch; cfg 5: 40 poses of 600-atom pockets) take the
    bench-sized path of the library (k_conv_layer: four convs of a layer in one k_conv grid + k_reduce_ln_layer); the same jobs
    in small batches take the small-batch path (persistent k_conv2 + tail split).  Per-job random streams + channel-owner
    summation in both kernels => the final poses must agree bit for bit -- a size-independent property that needs no oracle."""
    import bench
    dev = torch.device("cuda:0")
    _, _, _, samp = _hip(dev)
    jobs = bench.make_jobs(cfg_id, n_jobs, seed=9)
    big = ddist.run_sharded(samp, jobs, poses, seed=3, device=dev, batch_poses=n_jobs * poses)
    small = ddist.run_sharded(samp, jobs, poses, seed=3, device=dev, batch_poses=small_batch)
    for (l0, a0), (l1, a1) in zip(big, small):
        assert torch.isfinite(l0).all() and torch.equal(l0, l1) and torch.equal(a0, a1)


@pytest.mark.gpu
def test_host_store_equals_device_store_on_the_gpu():
    """store='host' (pose records streamed to pinned host memory batch by batch, bounded HBM) gives the very poses of store='device'."""
    dev = torch.device("cuda:0")
    _, _, _, samp = _hip(dev)
    _, jobs = make_jobs(3, 5, 60, 10, seed=6)
    poses = [3, 2, 7, 1, 2]
    a = ddist.run_sharded(samp, jobs, poses, seed=4, device=dev, batch_poses=4)
    b = ddist.run_sharded(samp, jobs, poses, seed=4, device=dev, batch_poses=4, store="host", gather="root", release=True)
    for (l0, a0), (l1, a1) in zip(a, b):
        assert l1.device.type == "cpu" and l1.is_pinned()
        assert torch.equal(l0.cpu(), l1) and torch.equal(a0.cpu(), a1)
    assert all("_dev" not in j.lig.__dict__ for j in jobs)          # halves released after their last batch
