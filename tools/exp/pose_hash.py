"""Developer: sha256 of the poses a few small and one batch-sized job list end on, + seconds per call -- run through two builds of the library
(DBFR_LIB=<path>) to show that a change left every bit where it was:   python tools/exp/pose_hash.py [cases...]"""
import hashlib
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import diffbindfr_amd as dba  # noqa: E402

CASES = {"cfg1": (1, 1, 4), "bs16": (2, 4, 4), "p160": (2, 4, 40), "c5p16": (5, 1, 16), "cfg1x40": (1, 1, 40), "bs64": (2, 16, 4)}
dev = torch.device("cuda:0")
model = bench.seeded_params().to(dev)
samp = dba.DiffBindFRHIP(diffusion_model=model, test_cfg={})
for case in (sys.argv[1:] or ["cfg1", "bs16", "p160", "c5p16"]):
    cfg_id, n_c, ppc = CASES[case]
    jobs = bench.make_jobs(cfg_id, n_c, seed=77)
    pb, lig, a14 = samp.run_complexes(jobs, ppc, dev, seed=1)
    torch.cuda.synchronize()
    h = hashlib.sha256(lig.cpu().numpy().tobytes() + a14.cpu().numpy().tobytes()).hexdigest()[:16]
    t0 = time.perf_counter()
    for r in range(3):
        samp.run_complexes(jobs, ppc, dev, seed=1)
    torch.cuda.synchronize()
    print(f"{case}: poses sha256 {h}  {(time.perf_counter() - t0) / 3:.4f} s per call  (lib: {os.environ.get('DBFR_LIB', 'tree')})")
