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
