"""Developer script: small-batch latency of one dbfr_sample call (cfg 1 literally: 1 complex x 4 poses)."""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
import diffbindfr_amd as dba
from diffbindfr_amd import synthetic
from diffbindfr_amd.packing import PackedBatch
dev = torch.device("cuda:0")
model = bench.seeded_params().to(dev)
samp = dba.DiffBindFRHIP(diffusion_model=model, test_cfg={})
recs, _ = samp.schedule()
for cfg_id, nc, np_ in ((1, 1, 4), (2, 1, 16), (2, 1, 40), (2, 2, 40), (2, 4, 40)):
    d = synthetic.make_batch(cfg_id, n_complex=nc, poses=np_, seed=1)
    for k, v in vars(d).items():
        if torch.is_tensor(v): setattr(d, k, v.to(dev))
    t0 = time.perf_counter(); pb = PackedBatch(d, dev); torch.cuda.synchronize(); t_pack = time.perf_counter() - t0
    z = {"tr": torch.randn(20, pb.G, 3, device=dev), "rot": torch.randn(20, pb.G, 3, device=dev),
         "tor": torch.randn(20, max(pb.dims["NTOR"], 1), device=dev), "sc": torch.randn(20, max(pb.dims["NSC"], 1), device=dev)}
    init = {k: pb.t[k].clone() for k in ("lig_pos", "rec_pos", "torsion_angle")}
    ts = []
    for it in range(4):
        for k, v in init.items(): pb.t[k].copy_(v)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        samp.sample_packed(pb, z)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"cfg{cfg_id} {nc}x{np_} poses: pack {t_pack*1e3:.1f} ms, sample {min(ts)*1e3:.1f} ms -> {np_*nc/min(ts):.1f} poses/s", flush=True)

# ---- f1: batch preparation for 16 complexes x 40 poses -- reference-style per-pose host transforms + collate (the
# oracle's restatement, CPU) vs ComplexRecord/assemble/init_poses (device)
import copy
import numpy as np
from oracle import pose_init as opi
from diffbindfr_amd import assemble
T = synthetic.residue_tables()
Tt = {k: torch.from_numpy(np.asarray(v)) if k == "atom14_to_group" else v for k, v in T.items()}
rng = np.random.default_rng(0)
recs = [synthetic.make_record(synthetic.make_pocket(rng, 200), synthetic.make_ligand(rng, 30), rng) for _ in range(16)]
t0 = time.perf_counter()
poses = []
for r in recs:
    fixed = opi.sc_fixer(copy.deepcopy(r), T)
    for _ in range(40):
        nt = int(r["tor_edge_mask"].sum())
        tp = dict(tor=rng.uniform(-np.pi, np.pi, nt), rot=np.eye(3), tr=torch.randn(1, 3) * 10, sc=rng.uniform(-np.pi, np.pi, (r["sequence"].shape[0], 4)))
        poses.append(opi.init_pose(copy.deepcopy(fixed), tp, Tt))
coll = opi.collate(poses)
pb_ref = PackedBatch({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in coll.items()}, dev)
torch.cuda.synchronize(); t_ref = time.perf_counter() - t0
t0 = time.perf_counter()
crs = [assemble.ComplexRecord(r) for r in recs]
t_rec = time.perf_counter() - t0
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pb = assemble.assemble(crs, 40, dev)
    assemble.init_poses(model, pb, assemble.draw_init_tape(pb))
    torch.cuda.synchronize(); t_new = time.perf_counter() - t0
print(f"f1 prep of 640 poses: per-pose host transforms + collate + pack {t_ref*1e3:.0f} ms | records once {t_rec*1e3:.1f} ms, "
      f"assemble + device init {t_new*1e3:.1f} ms", flush=True)
