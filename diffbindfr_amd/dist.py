"""Multi-GPU sharding of the (complex x pose) job list: one process per GPU, no
communication during sampling, one fixed-size gather at the end (RCCL over xGMI through
torch.distributed's "nccl" backend on ROCm; "gloo" on CPU for tests).

The reference offers only single-GPU inference + SLURM job arrays (README.md:129-150,
DiffBindFR/common/dataframe.py:203-234); its one multi-GPU inference gather is the
pickled-uint8 all_gather of druglib/core/runner/engine/test_utils.py:96-145.  Here the
records are fixed-size float tensors, so one ``all_gather_into_tensor`` suffices.
"""
import os
import sys

import torch
import torch.distributed as dist


def _single():
    """DBFR_DIST_SINGLE=1: developer switch -- form a process group and run every collective even with ONE rank, so that the multi-GPU
    path (RCCL calls included) can be walked end to end on a 1-GPU box: `DBFR_DIST_SINGLE=1 python bench.py`."""
    return os.environ.get("DBFR_DIST_SINGLE") == "1"


def _active():
    return dist.is_initialized() and (dist.get_world_size() > 1 or _single())


def init(backend=None):
    """Initialise from the torchrun environment (RANK/WORLD_SIZE/MASTER_*); no-op for 1 process (unless DBFR_DIST_SINGLE=1)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or _single()) and not dist.is_initialized():
        # dmabuf IPC is the only mode the host driver supports (RCCL's hipIpcGetMemHandle fails otherwise); must be in the
        # environment before the HSA runtime starts, i.e. before the first device call of this process
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # DBFR_DIST_BACKEND=gloo: developer override to walk the multi-rank path on a box with fewer GPUs than ranks
        backend = backend or os.environ.get("DBFR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":     # one rank per GPU: bind the communicator to this rank's device up front (no "guessing device ID" at the
            dev = torch.device("cuda", local % max(1, torch.cuda.device_count()))   # first barrier, errors surface here and not mid-run)
            torch.cuda.set_device(dev)
            kw["device_id"] = dev
        # an explicit rendezvous / collective timeout (default 10 minutes, DBFR_DIST_TIMEOUT_S): a rank that never arrives fails the job with a
        # message instead of leaving the others in a barrier for the library default of 30 minutes
        import datetime
        kw["timeout"] = datetime.timedelta(seconds=float(os.environ.get("DBFR_DIST_TIMEOUT_S", "600")))
        pin_rank_to_cores(local, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        if rank == 0:       # one line, so that an RCCL start-up failure can be told from a sampler failure in the driver's log
            ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
            print(f"[dbfr.dist] backend={backend} (nccl = RCCL on ROCm) world={world} devices_visible={ndev} "
                  f"master={os.environ['MASTER_ADDR']}:{os.environ['MASTER_PORT']} "
                  f"HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}", file=sys.stderr, flush=True)
    return rank, world, local


def pin_rank_to_cores(local_rank, local_world):
    """One rank per GPU shares the node's host cores with the other ranks: the host side of a batch (assemble: record halves ->
    packed batch, a few ms of numpy / torch-CPU work per batch) runs on this rank's own contiguous slice of the cores the process may
    use, so that eight ranks neither bounce between NUMA nodes nor oversubscribe one another's cores.  DBFR_DIST_PIN=0 switches it
    off; a launcher that already restricted the affinity to fewer cores than a slice is left alone.  Returns the cores or None."""
    if os.environ.get("DBFR_DIST_PIN", "1") == "0" or local_world <= 1 or not hasattr(os, "sched_getaffinity"):
        return None
    try:
        cores = sorted(os.sched_getaffinity(0))
        per = len(cores) // local_world
        if per < 1:
            return None
        mine = cores[local_rank * per:(local_rank + 1) * per]
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(per, int(os.environ.get("OMP_NUM_THREADS", str(per))))))
        return mine
    except OSError:
        return None


def complex_cost(n_atoms, n_lig, n_cab=None):
    """Edge-count cost model of one pose-step."""
    n_cab = n_atoms * 0.236 if n_cab is None else n_cab
    return 10.4 * n_atoms + 15 * n_lig + 2 * n_lig * (n_cab + 11)


def shard_lpt(costs, world, speeds=None):
    """Longest-processing-time-first assignment of complexes (all poses of a complex stay on
    one GPU so its static tensors are uploaded once).  Returns list[world] of index lists.
    ``speeds``: relative throughput of every rank (``rank_speeds``; None = all equal): a job goes to the rank that would FINISH it
    first, so a board that holds a higher clock at the power cap takes proportionally more of the table and all ranks end together."""
    speeds = [1.0] * world if speeds is None else [float(v) for v in speeds]
    assert len(speeds) == world and min(speeds) > 0
    order = sorted(range(len(costs)), key=lambda i: -costs[i])
    load = [0.0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: ((load[k] + costs[i]) / speeds[k], k))
        out[r].append(i)
        load[r] += costs[i]
    for o in out:
        o.sort()
    return out


def gather_records(local, world=None):
    """all_gather of equally-shaped per-rank record tensors [n, R] -> [world*n, R] on every rank."""
    if not _active():
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
    if not _active():
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
    if not _active():
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t)


def all_gather_vec(vals, device):
    """A few numbers per rank, from every rank to every rank: list[world] of lists (bench.py's per-rank read-out)."""
    if not _active():
        return [[float(v) for v in vals]]
    t = torch.tensor([float(v) for v in vals], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.tolist() for o in out]


def barrier():
    if _active():
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


def shard_jobs(jobs, poses, world, speeds=None):
    """LPT over the jobs' pose-step cost (``ComplexRecord.cost`` x poses); all poses of a job stay on one rank."""
    reps = [poses] * len(jobs) if isinstance(poses, int) else list(poses)
    return shard_lpt([j.cost * p for j, p in zip(jobs, reps)], world, speeds), reps


def rank_speeds(seconds, device, spread=0.10):
    """Relative speed of every rank from the time each took for the SAME calibration batch (all-gathered, so every rank derives the same
    list): 1 / t, normalised to mean 1, held within +-``spread`` of it (a mis-timed calibration must not unbalance the table by more
    than the boards can differ: MI355X boards measured here hold clocks 5 % apart at the power cap).  One rank: [1.0]."""
    ts = [v[0] for v in all_gather_vec([float(seconds)], device)]
    inv = [1.0 / max(t, 1e-9) for t in ts]
    mean = sum(inv) / len(inv)
    return [min(1.0 + spread, max(1.0 - spread, v / mean)) for v in inv]


def free_device_bytes(dev):
    """Free bytes on ``dev`` (None for the host)."""
    dev = torch.device(dev)
    return torch.cuda.mem_get_info(dev)[0] if dev.type == "cuda" else None


def release_needed(records, dev, share=0.25):
    """``run_sharded(release="auto")``: do the device copies of these records' halves (each shared half counted once) take more than
    ``share`` of the memory free on ``dev``?  Host-side arithmetic only."""
    free = free_device_bytes(dev)
    if free is None:
        return False
    halves = {id(h): h for r in records for h in (r.lig, r.pocket)}
    return sum(h.nbytes() for h in halves.values()) > share * free


def _gather_windows(local, n_all, world, rank, mode, window, stage_dev):
    """Bring the flat record buffers of all ranks together in bounded windows.

    ``local``: this rank's 1-D float buffer (device or pinned host); ``n_all[r]``: floats in use on rank r (every rank derives all
    of them from the job table, so no size exchange is needed and idle ranks walk the same windows).  ``mode``: "all" (every
    rank receives everything: ``all_gather_into_tensor``) or "root" (rank 0 only: ``gather``).  At most ``window`` floats per
    rank travel per collective, staged through ``stage_dev`` (the rank's GPU for RCCL, the host for gloo), so the device
    memory the gather needs is ``(world + 1) * window * 4`` bytes whatever the size of the job table.
    Returns list[world] of buffers like ``local`` (None for ranks this rank does not receive)."""
    n_valid, n_max = int(n_all[rank]), int(max(n_all))
    recv = mode == "all" or rank == 0
    like = dict(dtype=local.dtype, device=local.device, pin_memory=local.device.type == "cpu" and torch.cuda.is_available())
    out = [torch.empty(int(n_all[r]), **like) if recv and r != rank else None for r in range(world)]
    out[rank] = local[:n_valid]
    for w0 in range(0, n_max, window):
        w = min(window, n_max - w0)
        send = torch.zeros(w, dtype=local.dtype, device=stage_dev)
        k = max(0, min(w, n_valid - w0))
        if k:
            send[:k] = local[w0:w0 + k].to(stage_dev)          # blocking on purpose: a host-staged (gloo) collective reads it next
        if mode == "all":
            buf = torch.empty(world * w, dtype=local.dtype, device=stage_dev)
            dist.all_gather_into_tensor(buf, send)
            parts = buf.view(world, w)
        else:
            parts = [torch.empty(w, dtype=local.dtype, device=stage_dev) for _ in range(world)] if rank == 0 else None
            dist.gather(send, parts, dst=0)
        if recv:
            for r in range(world):
                k = max(0, min(w, int(n_all[r]) - w0))
                if r != rank and k:
                    out[r][w0:w0 + k] = parts[r][:k].to(out[r].device)
    return out


def run_sharded(sampler, jobs, poses, seed, device, batch_poses=640, tr_sigma_max=10.0, gather=True, on_batch=None,
                store="device", window_bytes=256 << 20, release="auto", tapes=None, rank_speed=None):
    """The multi-GPU product entry: a job list in, poses out in job order.

        jobs    list of ``assemble.ComplexRecord`` -- the (protein, ligand) pair table of the reference
                (DiffBindFR/common/dataframe.py:190-234).  Forward screen (BASELINE config 3): the records share ONE
                ``PocketRecord``; target fishing (config 4): they share ONE ``LigandRecord``; the shared half is
                replicated on every rank and uploaded once per device.
        poses   poses per job (int or list)
        gather  True / "all": every rank returns every job's poses (one ``all_gather_into_tensor`` per window);
                "root": only rank 0 does (``gather`` to rank 0 -- a forward screen's 10 k jobs x 40 poses are 1.8 GB: no
                reason to land them on all eight GPUs); False: no collective, every rank returns its own jobs.
        store   "device": the pose records stay in HBM; "host": every batch's records go to pinned host memory as the batch
                finishes and the gather is staged through a ``window_bytes`` device buffer -- HBM use is then bounded by one
                batch + the window, whatever the size of the job table.
        release True: drop a record half's device copy (``_Half.release``) after the last batch of this rank that uses it -- for job
                tables whose records do not fit HBM together (the forward screen's 10 k ligands do, at 20 KB each).  It MUTATES the
                caller's records (a second run uploads them again).  False: the records stay as they were uploaded.
                "auto" (default): True only if the device copies of this rank's records would take more than a quarter of the
                device memory that is free when the call starts (``release_needed``) -- never at the sizes of BASELINE.json.
        tapes   {job index: (init tape, noise tape)} of recorded random numbers used instead of drawing (``draw_tapes``).
        rank_speed  list[world] of relative rank speeds (``rank_speeds``: measured on a calibration batch), THE SAME LIST ON EVERY
                RANK: the LPT shard weighs the ranks by it.  The poses do not depend on it (per-job random streams).

    Every rank holds the whole (cheap, host-side) job table, takes its LPT share, runs it in batches of <= ``batch_poses``
    poses through ``run_complexes``.  Every JOB draws from its own generator ``job_seed(seed, job)``; a job cut into several
    batches draws its whole tape each time and uses its rows (``pose_ranges``), so the result depends neither on the
    sharding nor on ``batch_poses``.  Records are RAGGED: job j contributes ``poses_j x (3 N_l,j + 42 N_r,j)`` floats to its
    rank's flat buffer (no padding to the largest complex).  No collective on the data path before the final gather (RCCL
    over xGMI; the reference's counterpart is the pickled all_gather of druglib/core/runner/engine/test_utils.py:96-145).
    Returns list[len(jobs)] of (lig [P, N_l, 3], atom14 [P, N_r, 14, 3]) in job order (None for jobs this rank does not hold)."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    dev = torch.device(device)
    assert store in ("device", "host") and gather in (True, False, "all", "root")
    mode = "all" if gather is True else gather
    shards, reps = shard_jobs(jobs, poses, world, rank_speed)
    width = [3 * j.n_l + 42 * j.n_r for j in jobs]
    # flat record buffer of a rank: its jobs in shard order, poses consecutive
    off = [{} for _ in range(world)]
    n_flt = [0] * world
    for r in range(world):
        for j in shards[r]:
            off[r][j] = n_flt[r]
            n_flt[r] += reps[j] * width[j]
    on_host = store == "host" or dev.type == "cpu"
    local = (torch.empty(max(n_flt[rank], 1), pin_memory=dev.type == "cuda") if on_host
             else torch.empty(max(n_flt[rank], 1), device=dev))
    batches = plan_batches([(j, reps[j]) for j in shards[rank]], batch_poses)
    if release == "auto":
        release = release_needed([jobs[j] for j in shards[rank]], dev)
    last_use = {}
    for bi, batch in enumerate(batches):
        for j, _, _ in batch:
            last_use[id(jobs[j].lig)] = last_use[id(jobs[j].pocket)] = bi
    for bi, batch in enumerate(batches):
        recs = [jobs[j] for j, _, _ in batch]
        pb, lig, a14 = sampler.run_complexes(recs, [n for _, _, n in batch], device=dev, tr_sigma_max=tr_sigma_max,
                                             seeds=[sampler.job_seed(seed, j) for j, _, _ in batch],
                                             pose_ranges=[(p0, reps[j]) for j, p0, _ in batch],
                                             **({} if tapes is None else {"tapes": [tapes.get(j) for j, _, _ in batch]}))
        lp, rp = pb.lig_ptr_host.tolist(), pb.res_ptr_host.tolist()
        g = 0
        for j, p0, n in batch:          # the n poses of a job are consecutive, equal-sized graphs: two strided copies per job
            nl, nr = jobs[j].n_l, jobs[j].n_r
            rows = local[off[rank][j] + p0 * width[j]: off[rank][j] + (p0 + n) * width[j]].view(n, width[j])
            rows[:, :3 * nl].copy_(lig[-1, lp[g]:lp[g + n]].reshape(n, 3 * nl), non_blocking=True)
            rows[:, 3 * nl:].copy_(a14[-1, rp[g]:rp[g + n]].reshape(n, 42 * nr), non_blocking=True)
            g += n
        if on_host and dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()      # the batch's tensors are recycled by the next batch
        if release:
            for j, _, _ in batch:
                for half in (jobs[j].lig, jobs[j].pocket):
                    if last_use.get(id(half)) == bi:
                        half.release()
        if on_batch is not None:
            on_batch(bi, sum(n for _, _, n in batch))
    if _active() and mode:
        gloo = dist.get_backend() == "gloo"
        stage = torch.device("cpu") if gloo else dev
        bufs = _gather_windows(local, n_flt, world, rank, mode, max(1, int(window_bytes) // 4), stage)
    else:
        bufs = [local if r == rank else None for r in range(world)]
    res = [None] * len(jobs)
    for r in range(world):
        if bufs[r] is None:
            continue
        for j in shards[r]:
            nl, nr = jobs[j].n_l, jobs[j].n_r
            rows = bufs[r][off[r][j]: off[r][j] + reps[j] * width[j]].view(reps[j], width[j])
            res[j] = (rows[:, :3 * nl].reshape(reps[j], nl, 3), rows[:, 3 * nl:].reshape(reps[j], nr, 14, 3))
    return res
