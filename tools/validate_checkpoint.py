#!/usr/bin/env python
"""What has to be run ONCE on a machine that has the trained weights (and, for step 3, the reference's environment) -- neither is
available offline: `DiffBindFR/weights/diffbindfr_paper.pth` comes from Zenodo (10.5281/zenodo.10843568, /root/reference/README.md:70-71).

    python tools/validate_checkpoint.py --ckpt DiffBindFR/weights/diffbindfr_paper.pth [--reference /path/to/DiffBindFR] [--device cuda:0]

  1. load the checkpoint through the drop-in's loader contract (strict=True, the way DiffBindFR/app/predict.py:118-125 ->
     druglib/core/runner/checkpoint.py loads it): every key must find its parameter; the e3nn buffer keys land in the key sinks;
  2. pack it for the device (dbfr_model_create) and report the DYNAMIC RANGE of the radial-MLP weights -- per conv the spread of its
     lin.3 row maxima in bits -- and which convs the library packs with per-row factors (dbfr_model_rowscaled_convs: a
     tensor-product run whose rows lie more than 2^17 apart), with the row depth found per conv; on seeded weights the list is
     empty, what a trained checkpoint holds nobody has seen here;
  3. with the reference importable (RDKit etc.): `examples/forward` (3DBS x its 15 SDF ligands), 40 poses each, seed 888, through the
     reference's own dataset pipeline with `model.type=DiffBindFRHIP`, and the rate of poses with ligand RMSD < 2 A against the crystal
     pose next to the notebook's 37.5 % (15 of 40, /root/reference/notebooks/AF2_model_docking.ipynb).

Every step prints what it needs when it is missing and the script stops there with exit status 2.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def stop(msg):
    print("validate_checkpoint: " + msg)
    sys.exit(2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", default=os.path.join("DiffBindFR", "weights", "diffbindfr_paper.pth"))
    ap.add_argument("--reference", default=os.environ.get("DIFFBINDFR_ROOT", "/root/reference"))
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--poses", type=int, default=40)
    ap.add_argument("--seed", type=int, default=888)
    args = ap.parse_args()

    import numpy as np
    import torch
    import diffbindfr_amd as dba

    # ---- 1. the loader contract
    if not os.path.exists(args.ckpt):
        stop(f"step 1: checkpoint '{args.ckpt}' not found (Zenodo 10.5281/zenodo.10843568 -> DiffBindFR/weights/, README.md:70-71)")
    ck = torch.load(args.ckpt, map_location="cpu")
    sd = ck.get("state_dict", ck) if isinstance(ck, dict) else ck
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items() if not k.startswith("ema_")}
    pre = "diffusion_model."
    dm = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)} or sd
    model = dba.TensorProductModelHIP({})
    missing, unexpected = model.load_state_dict(dm, strict=False)
    unexpected = [k for k in unexpected if ".tp." not in k and "final_tp_tor" not in k]       # e3nn's buffers: absorbed by the key sinks
    print(f"step 1: {len(dm)} checkpoint tensors under '{pre}', missing {len(missing)}, unexpected {len(unexpected)}, "
          f"e3nn buffer keys absorbed {len(model.ignored_keys)}")
    if missing or unexpected:
        stop(f"step 1: strict load would fail: missing {list(missing)[:5]}, unexpected {unexpected[:5]}")

    # ---- 2. dynamic range of what goes onto the fp16 matrix instruction
    print("step 2: radial-MLP weights (lin.3: [W, 144]); row-max spread in bits = log2(largest row maximum / smallest non-zero row maximum)")
    for name, mod in model.named_modules():
        if not hasattr(mod, "fc") or not hasattr(mod, "tp"):
            continue
        w = mod.fc.lin[3].weight.detach()
        if w.shape[1] != 144:
            continue
        rm = w.abs().amax(dim=1)
        nz = rm[rm > 0]
        spread = float(torch.log2(nz.max() / nz.min())) if len(nz) else 0.0
        b = mod.fc.lin[3].bias.detach().abs().max()
        print(f"  {name:28s} |w| max {float(w.abs().max()):.3e}  row-max spread {spread:5.1f} bits  |bias| max {float(b):.3e}  "
              f"zero rows {int((rm == 0).sum())}")
    if not torch.cuda.is_available():
        stop("step 2: no ROCm device: the per-run verdict needs dbfr_model_create (there is no CPU path)")
    dev = torch.device(args.device)
    model = model.to(dev)
    rs, fb = model.rowscaled_convs(dev), model.fallback_convs(dev)
    print("step 2: convs packed with one power of two per ROW (a tensor-product run's rows more than 2^17 apart behind one factor: "
          "k_conv2h<.., ROWF>, ~7 % slower on those launches; include/dbfr.h: dbfr_model_rowscaled_convs), with the row depth found:")
    for name, depth in sorted(rs.items()):
        print(f"  {name:28s} deepest row 2^-{depth} of its run's largest")
    if not rs:
        print("  none")
    print(f"step 2: convs that leave the fp16 kernels for the three-bf16-piece one (a bias 2^48 above its row): {fb or 'none'}")

    # ---- 3. the notebook's experiment through the reference's own pipeline
    if not os.path.isdir(os.path.join(args.reference, "DiffBindFR")):
        stop(f"step 3: reference tree '{args.reference}' not found (--reference / $DIFFBINDFR_ROOT)")
    sys.path.insert(0, args.reference)
    try:
        import rdkit  # noqa: F401
        from DiffBindFR.common.inference_dataset import InferenceDataset  # noqa: F401
        from DiffBindFR.common import engines  # noqa: F401
    except Exception as e:
        stop(f"step 3: the reference's pipeline does not import here ({type(e).__name__}: {e}); needs its environment (env.yaml: rdkit, "
             f"e3nn==0.5.1, torch-cluster, torch-scatter, ...)")
    # the reference's own CLI with the drop-in selected: nothing of the reference is modified
    import subprocess
    ex = os.path.join(args.reference, "examples", "forward")
    out = os.path.abspath("validate_ckpt_out")
    cmd = [sys.executable, os.path.join(args.reference, "DiffBindFR", "app", "predict.py"), "-l", os.path.join(ex, "mols"),
           "-p", os.path.join(ex, "3dbs_protein.pdb"), "-o", out, "-np", str(args.poses), "-gpu", "0", "-cpu", "8", "-bs", "16", "-eval",
           "--seed", str(args.seed), "--cfg-options", "model.type=DiffBindFRHIP"]
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, args.reference, os.environ.get("PYTHONPATH", "")]),
               DIFFBINDFR_PLUGINS="diffbindfr_amd")
    print("step 3:", " ".join(cmd))
    r = subprocess.run(cmd, env=env)
    if r.returncode:
        stop(f"step 3: predict.py exited with {r.returncode}")
    import glob
    import pandas as pd
    frames = [pd.read_csv(f) for f in glob.glob(os.path.join(out, "**", "results_ec.csv"), recursive=True)] or \
             [pd.read_csv(f) for f in glob.glob(os.path.join(out, "**", "*.csv"), recursive=True)]
    col = next((c for f in frames for c in f.columns if "rmsd" in c.lower() and "sc" not in c.lower()), None)
    if not frames or col is None:
        stop(f"step 3: no ligand-RMSD column under {out}")
    rmsd = np.concatenate([f[col].to_numpy() for f in frames if col in f.columns])
    print(f"step 3: {len(rmsd)} poses, ligand RMSD < 2 A: {100.0 * float((rmsd < 2).mean()):.1f} %  (notebook, 3dbs x its crystal ligand, 40 poses: 37.5 %)")


if __name__ == "__main__":
    main()
