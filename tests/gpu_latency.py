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
