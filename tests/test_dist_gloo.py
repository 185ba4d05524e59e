"""world_size-2 gloo test of the N>1 path (sharding + the final gather) on CPU."""
import os
import socket

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffbindfr_amd import dist as ddist

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = ddist.init(backend="gloo")
    assert (r, w) == (rank, world)
    parts = ddist.shard_lpt([5.0, 1.0, 3.0, 2.0, 4.0], world)
    mine = parts[rank]
    local = torch.tensor([[float(i), float(rank)] for i in mine])
    allr = ddist.gather_ragged(local, len(mine))
    fixed = ddist.gather_records(torch.full((2, 3), float(rank)))
    mx = ddist.max_over_ranks(1.0 + rank, "cpu")
    vec = ddist.all_gather_vec([rank, 10 + rank, 0.5 * rank], "cpu")          # bench.py's per-rank read-out
    sp = ddist.rank_speeds(2.0 if rank == 0 else 2.5, "cpu")                  # rank 0 took 2.0 s for the calibration batch, rank 1 2.5 s
    ddist.barrier()
    q.put((rank, allr.tolist(), fixed.tolist(), mx, vec, sp))
    dist.destroy_process_group()


def test_gather_over_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, a0, f0, m0, v0, s0), (r1, a1, f1, m1, v1, s1) = res
    assert s0 == s1 and abs(s0[0] - 1.1) < 1e-9 and abs(s0[1] - 0.9) < 1e-9   # same list on both ranks; 1/2.0 : 1/2.5 = 1.111 : 0.889, held within +-10 %
    assert a0 == a1 and f0 == f1 and m0 == m1 == 2.0
    assert v0 == v1 == [[0.0, 10.0, 0.0], [1.0, 11.0, 0.5]]            # every rank sees every rank's numbers, in rank order
    assert sorted(int(x[0]) for x in a0) == [0, 1, 2, 3, 4]            # every complex gathered exactly once
    assert f0 == [[0.0] * 3] * 2 + [[1.0] * 3] * 2


def test_ranks_pin_themselves_to_disjoint_core_slices():
    """dist.pin_rank_to_cores (8-GPU insurance): eight local ranks take eight disjoint, equally long slices of the cores this process may use;
    run in child processes so that the test process keeps its own affinity."""
    import json
    import subprocess
    import sys
    if not hasattr(os, "sched_getaffinity") or len(os.sched_getaffinity(0)) < 8:
        pytest.skip("needs sched_setaffinity and at least eight cores")
    code = ("import json, os, sys; sys.path.insert(0, %r); from diffbindfr_amd import dist as d; "
            "print(json.dumps([d.pin_rank_to_cores(int(sys.argv[1]), 8), sorted(os.sched_getaffinity(0))]))" % ROOT)
    got = [json.loads(subprocess.run([sys.executable, "-c", code, str(r)], capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1]) for r in (0, 3, 7)]
    for mine, aff in got:
        assert mine == aff and len(mine) == len(os.sched_getaffinity(0)) // 8
    assert not (set(got[0][0]) & set(got[1][0])) and not (set(got[1][0]) & set(got[2][0]))
