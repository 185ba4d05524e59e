"""Multi-GPU sharding of the (complex x pose) job list: one process per GPU, no
communication during sampling, one fixed-size gather at the end (RCCL over xGMI through
torch.distributed's "nccl" backend on ROCm; "gloo" on CPU for tests).

The reference offers only single-GPU inference + SLURM job arrays (README.md:129-150,
DiffBindFR/common/dataframe.py:203-234); its one multi-GPU inference gather is the
pickled-uint8 all_gather of druglib/core/runner/engine/test_utils.py:96-145.  Here the
records are fixed-size float tensors, so one ``all_gather_into_tensor`` suffices.
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise from the torchrun environment (RANK/WORLD_SIZE/MASTER_*); no-op for 1 process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        # dmabuf IPC is the only mode the host driver supports (RCCL's hipIpcGetMemHandle fails otherwise); must be in the
        # environment before the HSA runtime starts, i.e. before the first device call of this process
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # DBFR_DIST_BACKEND=gloo: developer override to walk the multi-rank path on a box with fewer GPUs than ranks
        backend = backend or os.environ.get("DBFR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def complex_cost(n_atoms, n_lig, n_cab=None):
    """Edge-count cost model of one pose-step."""
    n_cab = n_atoms * 0.236 if n_cab is None else n_cab
    return 10.4 * n_atoms + 15 * n_lig + 2 * n_lig * (n_cab + 11)


def shard_lpt(costs, world):
    """Longest-processing-time-first assignment of complexes (all poses of a complex stay on
    one GPU so its static tensors are uploaded once).  Returns list[world] of index lists."""
    order = sorted(range(len(costs)), key=lambda i: -costs[i])
    load = [0.0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += costs[i]
    for o in out:
        o.sort()
    return out


def gather_records(local, world=None):
    """all_gather of equally-shaped per-rank record tensors [n, R] -> [world*n, R] on every rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = world or dist.get_world_size()
    dev = local.device
    if dist.get_backend() == "gloo" and local.is_cuda:      # gloo has no device collectives: stage through the host
        local = local.cpu()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out.to(dev)


def gather_ragged(local, n_valid):
    """Gather record tensors whose leading dim differs per rank: pad to the max, gather, trim."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local[:n_valid]
    world = dist.get_world_size()
    n = torch.tensor([n_valid], dtype=torch.int64, device="cpu" if dist.get_backend() == "gloo" else local.device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n)
    nmax = int(max(int(x) for x in ns))
    pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:n_valid] = local[:n_valid]
    allr = gather_records(pad, world)
    return torch.cat([allr[r * nmax:r * nmax + int(ns[r])] for r in range(world)], 0)


def max_over_ranks(x, device):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


# ------------------------------------------------------------------------------------------------ the sharded job driver
def plan_batches(job_poses, batch_poses):
    """Cut a rank's job list into device batches: jobs stay whole (their static tensors are assembled once per batch), a
    job with more poses than ``batch_poses`` is cut into chunks at multiples of ``batch_poses``, and the jobs are spread
    over ceil(total / batch_poses) batches of near-equal pose count (no tiny last batch).
    ``job_poses``: list of (job index, poses).  Returns list of batches = list of (job index, first pose, n poses)."""
    units = []
    for j, p in job_poses:
        for p0 in range(0, p, batch_poses):
            units.append((j, p0, min(batch_poses, p - p0)))
    total = sum(u[2] for u in units)
    if total == 0:
        return []
    n_b = max(1, -(-total // batch_poses))
    target = total / n_b
    out, cur, acc, done = [], [], 0, 0
    for u in units:
        # close the current batch when adding this unit would overshoot the running target by more than stopping short
        if cur and (acc + u[2] > batch_poses or (len(out) < n_b - 1 and done + acc + u[2] / 2.0 > target * (len(out) + 1))):
            out.append(cur); done += acc; cur, acc = [], 0
        cur.append(u); acc += u[2]
    if cur:
        out.append(cur)
    return out


def shard_jobs(jobs, poses, world):
    """LPT over the jobs' pose-step cost (``ComplexRecord.cost`` x poses); all poses of a job stay on one rank."""
    reps = [poses] * len(jobs) if isinstance(poses, int) else list(poses)
    return shard_lpt([j.cost * p for j, p in zip(jobs, reps)], world), reps


def run_sharded(sampler, jobs, poses, seed, device, batch_poses=640, tr_sigma_max=10.0, gather=True, on_batch=None):
    """The multi-GPU product entry: a job list in, poses out in job order.

        jobs    list of ``assemble.ComplexRecord`` -- the (protein, ligand) pair table of the reference
                (DiffBindFR/common/dataframe.py:190-234).  Forward screen (BASELINE config 3): the records share ONE
                ``PocketRecord``; target fishing (config 4): they share ONE ``LigandRecord``; the shared half is
                replicated on every rank and uploaded once per device.
        poses   poses per job (int or list)

    Every rank holds the whole (cheap, host-side) job table, takes its LPT share, runs it in batches of <= ``batch_poses``
    poses through ``sample_complexes`` (per-job random streams => the result does not depend on the sharding), writes
    fixed-size pose records [ligand N_l,max x 3 | atom14 N_r,max x 14 x 3] and ONE ``all_gather_into_tensor`` brings them to
    every rank (RCCL over xGMI; the reference's counterpart is the pickled all_gather of
    druglib/core/runner/engine/test_utils.py:96-145).  No collective on the data path before that.
    Returns list[len(jobs)] of (lig [P, N_l, 3], atom14 [P, N_r, 14, 3]) device tensors in job order (None when
    ``gather=False`` on ranks that did not run the job)."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    dev = torch.device(device)
    shards, reps = shard_jobs(jobs, poses, world)
    max_nl = max(j.n_l for j in jobs)
    max_nr = max(j.n_r for j in jobs)
    R = 3 * max_nl + 42 * max_nr
    # record rows of a rank: its jobs in shard order, poses consecutive
    row0 = [{} for _ in range(world)]
    n_rows = [0] * world
    for r in range(world):
        for j in shards[r]:
            row0[r][j] = n_rows[r]
            n_rows[r] += reps[j]
    n_max = max(n_rows)
    local = torch.zeros(max(n_max, 1), R, device=dev)
    for bi, batch in enumerate(plan_batches([(j, reps[j]) for j in shards[rank]], batch_poses)):
        recs = [jobs[j] for j, _, _ in batch]
        pb, lig, a14 = sampler.run_complexes(recs, [n for _, _, n in batch], device=dev, tr_sigma_max=tr_sigma_max,
                                             seeds=[sampler.job_seed(seed, j, p0) for j, p0, _ in batch])
        lp, rp = pb.lig_ptr_host.tolist(), pb.res_ptr_host.tolist()
        g = 0
        for j, p0, n in batch:          # the n poses of a job are consecutive, equal-sized graphs: two strided copies per job
            nl, nr = jobs[j].n_l, jobs[j].n_r
            rows = local[row0[rank][j] + p0: row0[rank][j] + p0 + n]
            rows[:, :3 * nl] = lig[-1, lp[g]:lp[g + n]].reshape(n, 3 * nl)
            rows[:, 3 * max_nl: 3 * max_nl + 42 * nr] = a14[-1, rp[g]:rp[g + n]].reshape(n, 42 * nr)
            g += n
        if on_batch is not None:
            on_batch(bi, sum(n for _, _, n in batch))
    if world > 1 and gather:
        allr = gather_records(local, world).view(world, max(n_max, 1), R)
    else:
        allr = local.view(1, max(n_max, 1), R) if world == 1 else None
    res = [None] * len(jobs)
    for r in range(world):
        if allr is None and r != rank:
            continue
        src = allr[r] if allr is not None else local
        for j in shards[r]:
            rows = src[row0[r][j]: row0[r][j] + reps[j]]
            nl, nr = jobs[j].n_l, jobs[j].n_r
            res[j] = (rows[:, :3 * nl].reshape(reps[j], nl, 3), rows[:, 3 * max_nl: 3 * max_nl + 42 * nr].reshape(reps[j], nr, 14, 3))
    return res
