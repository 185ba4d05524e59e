"""Developer run for profiling the small-batch path: BASELINE configs[0] literally (1 complex of the 3DBS shape x 4 poses
x 20 steps) or predict.py's -bs 16 of the cfg-2 shape, a few repetitions, wall time printed.
    rocprofv3 --kernel-trace --stats -d gpurun_out/lat -- python tools/latency_run.py --case cfg1"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import diffbindfr_amd as dba  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--case", default="cfg1")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg_id, n_c, ppc = {"cfg1": (1, 1, 4), "bs16": (2, 4, 4), "bs40": (2, 1, 40), "p80": (2, 2, 40), "p160": (2, 4, 40), "p320": (2, 8, 40),
                    "c5p16": (5, 1, 16), "cfg1x40": (1, 1, 40), "bs32": (2, 8, 4), "bs64": (2, 16, 4), "bs128": (2, 32, 4), "bs256": (2, 64, 4),
                    "p640": (2, 16, 40)}[a.case]
model = bench.seeded_params().to(dev)
samp = dba.DiffBindFRHIP(diffusion_model=model, test_cfg={})
jobs = bench.make_jobs(cfg_id, n_c, seed=77)
samp.run_complexes(jobs, ppc, dev, seed=1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for r in range(a.reps):
    samp.run_complexes(jobs, ppc, dev, seed=2 + r)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.reps
print(f"{a.case}: {n_c} complexes x {ppc} poses = {n_c * ppc} graphs per batch: {dt:.4f} s per batch, {n_c * ppc / dt:.1f} poses/s")
