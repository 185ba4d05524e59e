#!/usr/bin/env python
"""Benchmark of the MI355X-native DiffBindFR sampler (BASELINE.json metric: poses/sec).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5|1] [--batch-poses B] [--scaling weak|strong]
                    [--jobs J] [--store device|host] [--gather all|root]

`--gpus N` with N > 1 and no torchrun environment: the script spawns its own N ranks (one process per GPU, LOCAL_RANK -> cuda:LOCAL_RANK,
rendezvous on 127.0.0.1), relays rank 0's ONE JSON line and returns the ranks' worst exit code; under `python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N` it uses the environment it finds (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).

One "step" = one pass of the hot path over one device batch: B synthetic (complex x pose) graphs of the named
BASELINE.json config taken through all 20 reverse-diffusion steps by ONE dbfr_sample call (score network x20 + SDE
updates, everything on the device).  The batches are cut from a JOB LIST by the product's own multi-GPU driver
(diffbindfr_amd.dist.run_sharded): LPT shard over the ranks -> per-rank batches of <= B poses -> per-job random
streams -> ONE all_gather_into_tensor of the fixed-size pose records (ligand + atom14) over RCCL, results in job order.
  weak   (default) the job list holds K batches PER RANK (K x B x N poses): per-GPU work fixed;
  strong the job list holds K batches in total, whatever N.
The default K=8 x B=640 (16 complexes x 40 poses per batch) at N=1 is exactly the PoseBusters-shape batch of
BASELINE.json configs[1] (128 complexes x 40 poses = 5120 poses); with any K >= 8 the line also carries `value_literal_128x40`,
the rate over the first 5120 poses.  --config 3 / 4: every job shares ONE receptor /
ONE ligand record (forward screen / target fishing).  Per-complex records are resident in HBM before the timed region
(uploaded once per ligand / pocket); fp32 arithmetic throughout (the reference runs fp32).  Rank 0 prints ONE JSON line.
After the timed region (N = 1) the script measures the dominant kernel's HBM traffic itself: two child runs of one batch under
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (`--no-pmc` skips them), then the latency cases and the CPU-oracle baseline.
"""
import argparse
import copy
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import diffbindfr_amd as dba  # noqa: E402
from diffbindfr_amd import assemble, dist as ddist, lib as L, synthetic  # noqa: E402
from diffbindfr_amd.packing import PackedBatch  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E ~8 TB/s
FP32_MATRIX_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, exact fp32
PIPE_RF = dict(kernel="k_convz + k_conv2h (one launch pair per interaction layer / for the two torsion heads): the rows of lin.3 that feed a scalar output "
                      "irrep reduce-first (Z = sum over a target's edges of y (x) h, then the 144 x W GEMM once per target segment), the vector-output rows per edge",
               instruction="v_mfma_f32_16x16x32_f16", products=3, peak=2500.0, hidden_on_pipe=True, sustained=1937.0,
               arithmetic="three fp16 x fp16 partial products per fp32 product (two fp16 pieces per operand, exact power-of-two scalings), fp32 accumulation")
PMC_FILE = {"f32": "profiles/r2_pmc_k_conv.json",
            "split_f16": "profiles/r4_pmc_k_conv2h.json", "reduce_first": "profiles/r5_pmc_conv_pair.json"}
PMC_FILE_CFG5 = {"split_f16": "profiles/r4_cfg5_pmc_k_conv2h.json", "reduce_first": "profiles/r5_cfg5_pmc_conv_pair.json"}
HALF_MATRIX_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA
W2_SHARE = 34992.0 / (34992.0 + 6 * 144.0)   # share of the conv flops 2*144*(144+W) that is the 144 x W GEMM (W = 2880, 3888, 4896, 7776 x3)
# what the matrix pipe executes per algorithmic fp32 product of the 144 x W GEMM, and on which instruction (include/dbfr.h: DBFR_GEMM_*)
PIPE = {
    "f32": dict(kernel="k_conv_layer<3> = the four k_conv<144> convs of an interaction layer as one grid (fused radial MLP + tensor product) + the "
                       "two k_conv<144> torsion-head convs; one 'launch' = one such grid",
                instruction="v_mfma_f32_16x16x4_f32", products=1, peak=FP32_MATRIX_PEAK_TFLOPS,
                arithmetic="v_mfma_f32_16x16x4_f32 (fp32 operands, bit-exact an fmaf chain)"),
    "split_f16": dict(kernel="k_conv2h (persistent; all four convs of an interaction layer, or the two torsion-head convs, per launch): fused radial MLP "
                             "+ tensor product; W2 pieces through an LDS ring, one copy per tile per CU, two barriers per tile",
                      instruction="v_mfma_f32_16x16x32_f16 (+ _16x16x16_f16 for the last 16 k)", products=3, peak=HALF_MATRIX_PEAK_TFLOPS,
                      hidden_on_pipe=True,
                      # profiles/r3_mfma_power.txt: what a bare stream of this instruction sustains on this board with random operands
                      # (the firmware's power management holds it at 2.05 GHz); the nominal peak is reached with all-zero operands only
                      sustained=1937.0,
                      arithmetic="three fp16 x fp16 partial products per fp32 product (operands scaled by exact powers of two and cut into two fp16 "
                                 "pieces: 23 of 24 significand bits), small and large products in separate fp32 accumulators -- error vs fp64 <= the "
                                 "fp32 MFMA's (max and rms)"),
}


def seeded_params(seed=1):
    """Seeded random weights of the reference architecture (no checkpoint offline), generated by
    the product's own module tree so bench.py does not depend on the oracle for the GPU leg."""
    model = dba.TensorProductModelHIP({})
    g = torch.Generator().manual_seed(seed)
    sd = model.state_dict()
    for k, v in sd.items():
        if "distance_expansion" in k:
            continue
        if k.endswith("mean_shift") or k.endswith("affine_weight"):
            v.add_(0.1 * torch.randn(v.shape, generator=g))
        elif k.endswith("affine_bias"):
            v.copy_(0.1 * torch.randn(v.shape, generator=g))
        elif "atom_emb_list" in k:
            v.copy_(0.5 * torch.randn(v.shape, generator=g))
        elif k.endswith(".weight"):
            v.copy_((torch.rand(v.shape, generator=g) * 2 - 1) / np.sqrt(v.shape[1]))
        elif k.endswith(".bias"):
            v.copy_((torch.rand(v.shape, generator=g) * 2 - 1) * 0.1)
    model.load_state_dict(sd)
    return model


def make_jobs(cfg_id, n_jobs, seed=0):
    """Synthetic job table of the named config: list of ComplexRecord.  cfg 3: ONE PocketRecord shared by all jobs;
    cfg 4: ONE LigandRecord shared.  Deterministic in (cfg, seed, job index): every rank builds the same table."""
    c = synthetic.CONFIGS[cfg_id]
    shared = c.get("shared")
    srng = np.random.default_rng(cfg_id * 1000 + 999983 + seed)
    sp = synthetic.make_pocket(srng, c["n_atoms"]) if shared == "receptor" else None
    sl = synthetic.make_ligand(srng, c["n_lig"]) if shared == "ligand" else None
    jobs, pocket0, lig0 = [], None, None
    for j in range(n_jobs):
        rng = np.random.default_rng([cfg_id * 1000 + seed, j])
        # +-10% size jitter around the named sizes so batches are ragged like real data
        pk = sp or synthetic.make_pocket(rng, int(round(c["n_atoms"] * rng.uniform(0.9, 1.1))))
        lg = sl or synthetic.make_ligand(rng, max(4, int(round(c["n_lig"] * rng.uniform(0.85, 1.15)))))
        raw = synthetic.make_record(pk, lg, np.random.default_rng([cfg_id, seed, 0 if shared else j, 17]))
        rec = assemble.ComplexRecord(raw, lig=lig0 if shared == "ligand" else None, pocket=pocket0 if shared == "receptor" else None)
        if j == 0:
            pocket0, lig0 = rec.pocket, rec.lig
        jobs.append(rec)
    return jobs


def cpu_baseline(cfg_id, samp, dev, n_poses=1, batched=(2, 2), batched_steps=5):
    """The oracle (op-for-op PyTorch-CPU restatement of the reference path) timed on the host cores on a bounded sample:
      (a) 1 complex x n_poses poses x all 20 steps  (cfg 1: its 4 poses) -- also the parity check: the same inputs and
          noise tape go through the HIP sampler, `parity` = deviation (A) of the two final poses;
      (b) cfg-shape batch of batched[0] complexes x batched[1] poses in ONE batched call for `batched_steps` of the 20
          steps, extrapolated linearly in steps (every step costs the same: same graphs, same network).
    `value` = the better of the two per-pose rates (the batched call amortises the per-op overheads of a G=1 batch)."""
    from oracle import sampler as osampler, schedule as osched, score_model as sm
    mcfg = sm.default_cfg()
    params = sm.init_params(mcfg, seed=1)
    scfg = osched.default_sample_cfg()
    T = synthetic.residue_tables()
    a14g = torch.from_numpy(T["atom14_to_group"]).long()
    for i in range(scfg.actual_steps):
        osched.step_scalars(scfg, i)          # warm the table caches outside the timed region
    threads = torch.get_num_threads()

    def run(n_complex, poses, steps, seed):
        d = synthetic.make_batch(cfg_id, n_complex=n_complex, poses=poses, seed=seed)
        G = d.num_graphs
        noise = osampler.draw_noise(scfg.actual_steps, G, int(d.tor_edge_mask.sum()), int(d.sc_torsion_edge_mask.sum()), 1)
        d_in = copy.deepcopy(d)
        cfg_s = scfg if steps == scfg.actual_steps else osched.default_sample_cfg(actual_steps=steps)
        t0 = time.time()
        lig_ref, a14_ref = osampler.sample(params, mcfg, cfg_s, d, noise, a14g)
        return G, time.time() - t0, d_in, noise, lig_ref, a14_ref

    G1, dt1, d_in, noise, lig_ref, a14_ref = run(1, n_poses, scfg.actual_steps, 4242)
    rate1 = G1 / dt1
    out = {"unit": "poses/s", "cores": threads, "kind": "port",
           "single": {"graphs": G1, "seconds": round(dt1, 1), "poses_per_sec": round(rate1, 5)}}
    sample = (f"cfg{cfg_id}-shape, torch CPU fp32 on {threads} threads (torch.get_num_threads()), materialised [E,W] weights, python "
              f"update loops as in the reference: (a) 1 complex x {G1} poses x {scfg.actual_steps} steps in {dt1:.1f}s")
    rate = rate1
    if batched and batched_steps > 0:
        Gb, dtb, *_ = run(batched[0], batched[1], batched_steps, 4343)
        full = dtb * scfg.actual_steps / batched_steps
        rateb = Gb / full
        out["batched"] = {"graphs": Gb, "steps_timed": batched_steps, "seconds": round(dtb, 1),
                          "seconds_extrapolated_20_steps": round(full, 1), "poses_per_sec": round(rateb, 5),
                          "per_pose_speedup_vs_single": round(rateb / rate1, 3)}
        sample += (f"; (b) {batched[0]} complexes x {batched[1]} poses in one batched call, {batched_steps} of {scfg.actual_steps} steps "
                   f"in {dtb:.1f}s, extrapolated linearly in steps to {full:.1f}s")
        rate = max(rate1, rateb)
    out["value"] = round(rate, 5)
    out["sample"] = sample
    # identical weights (the oracle's seeded init) + inputs + noise through the HIP path
    hip_model = dba.TensorProductModelHIP({}).to(dev)
    hip_model.load_state_dict(params, strict=True)
    hs = dba.DiffBindFRHIP(diffusion_model=hip_model, test_cfg={})
    for k, v in vars(d_in).items():
        if torch.is_tensor(v):
            setattr(d_in, k, v.to(dev))
    pb = PackedBatch(d_in, dev)
    z = {"tr": noise.tr, "rot": noise.rot, "tor": noise.tor, "sc": noise.sc}
    z = {k: (v if v.shape[1] else torch.zeros(v.shape[0], 1)).to(dev).contiguous() for k, v in z.items()}
    lig, a14 = hs.sample_packed(pb, z)
    dl = (lig[0].cpu() - lig_ref[0]).norm(dim=-1)
    da = (a14[0].cpu() - a14_ref[0]).norm(dim=-1)
    out["parity"] = {"what": "HIP vs CPU oracle, same inputs/weights/noise, after 20 steps",
                     "lig_rmsd_A": round(float((dl ** 2).mean().sqrt()), 7), "lig_max_dev_A": round(float(dl.max()), 7),
                     "pocket_max_dev_A": round(float(da.max()), 7), "tolerance_A": 1e-3}
    hip_model.release()
    return out


PMC_KERNEL = {"split_f16": ("k_conv2h<",), "reduce_first": ("k_convz<", "k_conv2h<")}      # matched on "name<": `k_conv` alone also matched final_conv's k_conv<96> and k_conv2


def measure_traffic(mode, args):
    """HBM traffic of the dominant kernel measured ON THIS BOX: two child runs of this script (one batch each) under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, counters only, as MI355X_MICROARCH.md prescribes (the two
    do not fit one pass) -- summed over the kernel's dispatches, per launch.  FETCH_SIZE is doubled (gfx950 tallies the 128-byte
    requests of 16-byte-per-lane reads at 64 bytes; same guide).  Returns (dict, None) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    kern = PMC_KERNEL.get(mode)
    if any("rocprof" in v.lower() for k, v in os.environ.items() if k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES")):
        return None, "no live PMC pass (this process already runs under a profiler)"
    if kern is None or shutil.which("rocprofv3") is None:
        return None, "no live PMC pass (rocprofv3 absent or fp32-instruction mode)"
    out = {}
    tmp = tempfile.mkdtemp(prefix="dbfr_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = ["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "1", "--warmup", "0", "--config", str(args.config), "--batch-poses", str(args.batch_poses),
                   "--no-profile", "--no-cpu-baseline", "--no-latency", "--no-native", "--no-pmc"]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=300)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {ctr} failed (rc {r.returncode})"
            tot, disp = 0.0, set()
            for f in files:
                for row in csv.DictReader(open(f)):
                    if any(k in row["Kernel_Name"] for k in kern) and row["Counter_Name"] == ctr:
                        tot += float(row["Counter_Value"])
                        # reduce_first: one 'launch' of the library's profile = k_convz + (where the conv has vector-output rows) k_conv2h;
                        # both kernels' bytes count, the launches are k_convz's
                        if mode != "reduce_first" or "k_convz<" in row["Kernel_Name"]:
                            disp.add(row["Dispatch_Id"])
            if not disp:
                return None, f"no {kern} dispatch in the {ctr} pass"
            out[ctr] = tot * 1024.0 / len(disp)          # the counters are in KiB
            out["dispatches"] = len(disp)
    except Exception as e:      # a profiler problem must not cost the bench line
        return None, f"live PMC pass failed: {type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out, None


class BoardSampler:
    """Board power and shader clock of one GPU while the timed region runs, read twice a second from the amdgpu hwmon files
    (/sys/class/drm/card*/device/hwmon/hwmon*/power1_input [uW], freq1_input [Hz], power1_cap) by a host thread -- two small file reads,
    no subprocess, nothing launched on the GPU.  The conv kernel runs at the board's power cap, so the clock the firmware grants differs
    from board to board: the line carries what THIS board did (`config.board`).  The card is found by the HIP device's PCI address.  Best effort: without the files it returns None and costs nothing."""

    def __init__(self, idx):
        import glob
        import threading
        # the card whose PCI address is the HIP device's (a node's other boards may show up in sysfs too)
        self.dir = None
        try:
            pr = torch.cuda.get_device_properties(idx)
            want = "%04x:%02x:%02x." % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
                if os.path.basename(os.path.realpath(f.split("/hwmon/")[0])).startswith(want):
                    self.dir = os.path.dirname(f)
                    break
        except Exception:
            pass
        self.rows = []
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True) if self.dir else None
        # which limiter holds the clock: the firmware's violation accumulators (amdsmi_get_violation_status through the amdsmi package ROCm ships,
        # tools/board_limiter.py), read once before and once after the timed region -- two library calls, nothing launched on the GPU
        self.lim, self.v0 = None, None
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import board_limiter
            self.lim = board_limiter.Limiter(0, bdf=want if self.dir else None)
            self._bl = board_limiter
        except Exception:
            self.lim = None

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name)) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return None

    def _run(self):
        while not self._stop.wait(0.5):
            pw, ck = self._read("power1_input"), self._read("freq1_input")
            if pw is not None or ck is not None:
                self.rows.append((pw / 1e6 if pw is not None else None, ck / 1e6 if ck is not None else None))

    def start(self):
        if self.lim is not None:
            self.v0 = self.lim.violation()
        if self._th:
            self._th.start()
        return self

    def stop(self):
        if not self._th:
            return None
        self._stop.set()
        self._th.join(timeout=5)
        rows = self.rows[4:] if len(self.rows) > 8 else self.rows       # (the first two seconds ramp up)
        if not rows:
            return None
        pw = [r[0] for r in rows if r[0] is not None]
        ck = [r[1] for r in rows if r[1] is not None]
        cap = self._read("power1_cap")
        limiter = None
        if self.lim is not None:
            try:
                limiter = self._bl.accumulator_shares(self.v0, self.lim.violation())
                if limiter:
                    limiter = {k: v for k, v in limiter.items() if not k.startswith("delta_") or k in ("delta_ppt_pwr",)}
                    limiter["what"] = ("share of the power-management firmware's control iterations during the timed region in which each controller held the "
                                       "clock down (amdsmi_get_violation_status accumulators, after - before): ppt_pwr = package power tracking, socket / vr / hbm "
                                       "_thrm = thermal, prochot")
                self.lim.close()
            except Exception:
                limiter = None
        return {"power_cap_w": cap / 1e6 if cap else None, "samples": len(rows), "limiter": limiter,
                "power_w_mean": round(sum(pw) / len(pw), 1) if pw else None, "power_w_max": round(max(pw), 1) if pw else None,
                "sclk_mhz_mean": round(sum(ck) / len(ck), 1) if ck else None, "sclk_mhz_min": round(min(ck), 1) if ck else None,
                "source": f"{self.dir}/power1_input, freq1_input, twice a second during the timed region (rank 0's board)"}


def latency_leg(samp, dev, lib, model):
    """BASELINE configs[0] literally, as predict.py is used: ONE complex of the 3DBS shape x 4 poses x 20 steps
    (records resident, assemble + init + sample + status sync timed), predict.py's `-bs 16` of the cfg-2 shape, and the shape of the
    ONE sampler timing the reference publishes (notebooks/AF2_model_docking.ipynb, 1 complex x 40 poses x 20 steps
    in 76.1 s = 0.53 poses/s on an unnamed CUDA GPU with trained weights -- another pocket, another GPU: context, not a baseline)."""
    out = {}
    for name, cfg_id, n_c, ppc in (("cfg1_1x4", 1, 1, 4), ("cfg2_bs16", 2, 4, 4), ("cfg1_1x40_notebook_shape", 1, 1, 40)):
        jobs = make_jobs(cfg_id, n_c, seed=77)
        for j in jobs:
            j.lig.dev(dev), j.pocket.dev(dev)
        samp.run_complexes(jobs, ppc, dev, seed=1)           # warm-up (workspace, streams)
        torch.cuda.synchronize(dev)
        lib.dbfr_profile_read(model.handle(dev), None, None, None, None, 1)
        lib.dbfr_profile_enable(model.handle(dev), 2)         # 2 = count flops only (no per-launch events: streams overlap)
        reps = 3
        t0 = time.perf_counter()
        for r in range(reps):
            samp.run_complexes(jobs, ppc, dev, seed=2 + r)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / reps
        fl = C.c_double()
        L.check(lib.dbfr_profile_read(model.handle(dev), None, None, C.byref(fl), None, 1))
        lib.dbfr_profile_enable(model.handle(dev), 0)
        tf = fl.value / reps / dt / 1e12
        out[name] = {"poses": n_c * ppc, "seconds": round(dt, 4), "poses_per_sec": round(n_c * ppc / dt, 2),
                     "conv_tflops_over_wall": round(tf, 2), "frac_of_fp32_matrix_peak": round(tf / FP32_MATRIX_PEAK_TFLOPS, 4)}
    out["cfg1_1x40_notebook_shape"]["reference_notebook"] = ("76.1 s = 0.53 poses/s for 1 complex x 40 poses x 20 steps on an unnamed CUDA GPU "
                                                             "(notebooks/AF2_model_docking.ipynb:274-281; BASELINE.md section 1) -- context only")
    return out


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: N children of this very command, one per GPU, with the torchrun environment
    (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR = 127.0.0.1, a free MASTER_PORT).  Rank 0's stdout carries the ONE JSON line and is
    relayed; the other ranks' stdout goes to stderr.  Any rank failing ends the others (by their exact pids) and is the exit code --
    the model is the reference's own multi-process test entry (druglib/core/runner/engine/test_utils.py:96-145 gathers from ranks a
    launcher started)."""
    import socket
    import subprocess
    import threading
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    base = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", LOCAL_WORLD_SIZE=str(n))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: the only mode the host driver supports (RCCL needs it)
    base.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    # the rendezvous port: a free one at this moment.  (The socket is closed right behind the Popen calls, seconds before rank 0 binds the port itself, so
    # a second bench started in that window CAN be handed the same number -- rank 0's bind then fails and this function ends all ranks with its error;
    # DBFR_DIST_TIMEOUT_S bounds the wait of the others.)
    sk = socket.socket()
    sk.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    sk.bind(("127.0.0.1", 0))
    base["MASTER_PORT"] = str(sk.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0", ROLE_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    sk.close()
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)      # rank 0's pipe is drained by a thread ...
    reader.start()
    rcs = [None] * n
    while any(c is None for c in rcs):      # ... while ALL children are watched: the first one to fail ends the others (exact pids) instead of
        for r, p in enumerate(procs):       # leaving them in a collective until its timeout
            if rcs[r] is None:
                rcs[r] = p.poll()
        if any(c not in (None, 0) for c in rcs):
            deadline = time.time() + 15
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    try:
                        rcs[r] = p.wait(timeout=max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        p.kill()
                        rcs[r] = p.wait()
            break
        time.sleep(0.2)
    reader.join(timeout=10)
    out0 = out0[0] if out0 else ""
    lines = [l for l in out0.splitlines() if l.startswith('{"metric"')]
    for l in out0.splitlines():
        if not l.startswith('{"metric"'):
            print(l, file=sys.stderr)
    rc = next((c for c in rcs if c != 0), 0)
    if rc == 0 and len(lines) != 1:
        print(f"bench.py: expected ONE result line from rank 0, got {len(lines)}", file=sys.stderr)
        rc = 1
    if lines:
        print(lines[-1], flush=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--batch-poses", type=int, default=640)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-native", action="store_true", help="skip the one-batch run of the fp32-matrix-instruction kernels")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="no child runs under rocprofv3 --pmc (roofline.traffic falls back to the constant of profiles/)")
    ap.add_argument("--cpu-poses", type=int, default=None)
    ap.add_argument("--cpu-batched-steps", type=int, default=5)
    ap.add_argument("--jobs", type=int, default=None, help="size of the job table (weak scaling: per rank); default --steps x (--batch-poses // poses per job)")
    ap.add_argument("--store", choices=("device", "host"), default=None, help="dist.run_sharded: pose records in HBM, or streamed to pinned host memory batch by batch "
                                                                             "(default: device; host for the screening configs 3 / 4 on more than one GPU)")
    ap.add_argument("--no-probe", action="store_true", help="do not measure what the bare matrix pipe sustains on this board (4 s; roofline.frac_of_sustained then uses round 3's constant)")
    ap.add_argument("--no-board", action="store_true", help="do not sample board power / clock (amdgpu hwmon files) during the timed region")
    ap.add_argument("--no-speed-shard", action="store_true", help="N > 1: shard the job table evenly instead of by the ranks' measured speed")
    ap.add_argument("--gather", choices=("all", "root"), default=None, help="dist.run_sharded: every rank receives every pose, or rank 0 only (default: all; root for "
                                                                          "the screening configs 3 / 4 on more than one GPU: 10 k ligands x 40 poses are written out by one rank)")
    args = ap.parse_args()
    screen = args.config in (3, 4) and args.gpus > 1
    if args.store is None:
        args.store = "host" if screen else "device"
    if args.gather is None:
        args.gather = "root" if screen else "all"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # no launcher: be one
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    # The contract is ONE JSON line on stdout.  Native libraries write there too -- RCCL prints its version banner to the C stdout of
    # rank 0, block-buffered, i.e. it would land BEHIND the line when the process exits -- so file descriptor 1 is pointed at stderr for
    # the whole run and the line goes to a private duplicate of the real stdout.
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank, world, local = ddist.init()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (a launcher's environment that does not match the command line)"
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device(f"cuda:{local % torch.cuda.device_count()}")   # one rank per GPU; the modulo only matters for the gloo dry run
    torch.cuda.set_device(dev)
    lib = L.load()
    cfg = synthetic.CONFIGS[args.config]
    ppc = cfg["poses"]
    B = args.batch_poses
    jobs_per_batch = max(1, B // ppc)

    model = seeded_params().to(dev)
    samp = dba.DiffBindFRHIP(diffusion_model=model, test_cfg={})
    recs, _ = samp.schedule()
    T = len(recs)
    n_jobs = (args.jobs if args.jobs else args.steps * jobs_per_batch) * (world if args.scaling == "weak" else 1)
    jobs = make_jobs(args.config, n_jobs, seed=1)
    warm = make_jobs(args.config, jobs_per_batch, seed=2)
    for i in range(args.warmup):
        samp.run_complexes(warm, ppc, dev, seed=i)
    torch.cuda.synchronize(dev)
    # N > 1: boards hold clocks up to 5 % apart at the power cap, and a statically sharded job takes as long as its slowest rank.  Every
    # rank times THREE more untimed batches of the same jobs (workspace and code already warm; the median counts), the times are all-gathered and the job table
    # is LPT-sharded by measured speed (dist.rank_speeds): faster boards take proportionally more jobs.  The poses do not depend on it.
    speeds, calib_s = None, None
    if world > 1 and args.warmup > 0 and not args.no_speed_shard:      # (with --warmup 0 the batch would time one-off set-up costs)
        cal = []
        for i in range(3):              # the median of three: one hiccup (a page fault, a clock dip) must not move a tenth of the job table
            ddist.barrier()
            tc = time.perf_counter()
            samp.run_complexes(warm, ppc, dev, seed=12345 + i)
            torch.cuda.synchronize(dev)
            cal.append(time.perf_counter() - tc)
        calib_s = sorted(cal)[1]
        speeds = ddist.rank_speeds(calib_s, dev)
    shards, reps = ddist.shard_jobs(jobs, ppc, world, speeds)
    t_up = time.perf_counter()
    for j in shards[rank]:                       # per-complex records resident in HBM before the timed region
        jobs[j].lig.dev(dev), jobs[j].pocket.dev(dev)
    torch.cuda.synchronize(dev)
    t_up = time.perf_counter() - t_up
    h = model.handle(dev)
    if not args.no_profile:
        lib.dbfr_profile_read(h, None, None, None, None, 1)
        lib.dbfr_profile_enable(h, 1)
    done = []
    regrown0 = model.regrown
    # host time spent in batch assembly (record halves -> packed batch, launches included) inside the timed region, for the read-out
    asm_s = [0.0]
    _assemble = assemble.assemble

    def timed_assemble(*a, **k):
        t = time.perf_counter()
        try:
            return _assemble(*a, **k)
        finally:
            asm_s[0] += time.perf_counter() - t
    assemble.assemble = timed_assemble
    torch.cuda.reset_peak_memory_stats(dev)
    board = BoardSampler(dev.index).start() if rank == 0 and not args.no_board else None
    ddist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    stamps = []

    def on_batch(i, n):
        done.append(n)
        stamps.append(time.perf_counter())      # every batch ends with a status sync on its stream: the stamp is a completion time
    res = ddist.run_sharded(samp, jobs, ppc, seed=100, device=dev, batch_poses=B, on_batch=on_batch, store=args.store, gather=args.gather,
                            release=False, rank_speed=speeds)      # (the records stay resident: they were uploaded before the timed region)
    torch.cuda.synchronize(dev)
    assemble.assemble = _assemble
    t_local = time.perf_counter() - t0          # this rank's own clock, sampling + gather
    ddist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = ddist.max_over_ranks(elapsed, dev)
    board = board.stop() if board is not None else None
    t_sampled = (stamps[-1] - t0) if stamps else 0.0
    # what every rank did, all-gathered: the line shows that the collective saw `world` ranks, and on which devices
    per_rank = ddist.all_gather_vec([rank, dev.index, float(sum(done)), len(done), t_local, t_sampled,
                                     torch.cuda.max_memory_allocated(dev) / 2 ** 30, t_up, asm_s[0],
                                     (len(warm) * ppc / calib_s) if calib_s else 0.0], dev)
    n_steps = max(int(v[3]) for v in per_rank) if args.jobs else args.steps      # --jobs: a step is still one batch; the slowest rank's count
    counters = (C.c_int64 * 8)()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    L.check(lib.dbfr_status_sync(C.c_void_p(model.workspace_of(dev).data_ptr()), stream, counters))
    mine = res if (args.gather == "all" or rank == 0) else [res[j] for j in shards[rank]]
    assert all(r is not None and torch.isfinite(r[0]).all() and torch.isfinite(r[1]).all() for r in mine), "non-finite pose coordinates"
    poses = sum(reps)
    if args.gather == "all" or rank == 0:
        assert sum(int(r[0].shape[0]) for r in res) == poses

    def read_roofline(elapsed_s, mode):
        """The dominant kernel's line from the library's own HIP events (recorded on the stream the kernel runs on)."""
        ms, nl, fl, rb = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        L.check(lib.dbfr_profile_read(h, C.byref(ms), C.byref(nl), C.byref(fl), C.byref(rb), 1))
        lib.dbfr_profile_enable(h, 0)
        if not nl.value:
            return None
        fb = C.c_double()
        L.check(lib.dbfr_profile_fused_bytes(h, C.byref(fb)))
        alg = fl.value / (ms.value * 1e-3) / 1e12            # algorithmic fp32 flops 2*144*(144+W) per edge / kernel time
        P = PIPE_RF if mode == "reduce_first" else PIPE[mode]
        # what the named matrix instruction executes: `products` MFMA-flops per algorithmic flop of the 144 x W GEMM (97.6 % of the
        # conv's flops); in the retired k_conv2r the 144 x 144 hidden layer stayed on the fp32 instruction inside the same kernel and was
        # not counted, k_conv2h runs it on the same three-product form (W1h tiles)
        ex = alg if mode == "f32" else P["products"] * (1.0 if P.get("hidden_on_pipe") else W2_SHARE) * alg
        exe, use, formb = C.c_double(), C.c_double(), C.c_double()
        L.check(lib.dbfr_profile_executed_flops(h, C.byref(exe)))
        L.check(lib.dbfr_profile_useful_flops(h, C.byref(use), C.byref(formb)))
        if mode == "reduce_first":
            # the reduce-first kernels do not execute products x the reference algorithm's flops: the library counts what the pipe executes (per-edge
            # kernel: 3 x 2 x 144 x (144 + vector-output rows) per edge; k_convz: the matrix instructions it issues x 16384)
            ex = exe.value / (ms.value * 1e-3) / 1e12
        traffic, tsrc, traw = None, "no PMC pass for this workload", None
        live, why = (None, "--no-pmc") if (args.no_pmc or world != 1 or mode != main_mode) else measure_traffic(mode, args)
        if live is not None:
            traffic = 2.0 * live["FETCH_SIZE"] + live["WRITE_SIZE"]
            traw = {"fetch_size_raw": live["FETCH_SIZE"], "write_size": live["WRITE_SIZE"], "dispatches": live["dispatches"]}
            tsrc = ("measured in this run: two child runs of this command (one batch) under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` "
                    "(separate passes, counters only), per launch of this kernel; traffic = 2 x FETCH_SIZE (gfx950 correction for 16-byte-per-"
                    "lane reads, MI355X_MICROARCH.md) + WRITE_SIZE")
        else:
            pmc_rel = PMC_FILE.get(mode) if args.config == 2 else PMC_FILE_CFG5.get(mode) if args.config == 5 else None
            pmc = os.path.join(ROOT, pmc_rel) if pmc_rel else None
            if pmc and os.path.exists(pmc) and args.batch_poses == 640:
                pj = json.load(open(pmc))     # separate rocprofv3 --pmc passes (tools/prof_r3.sh), NOT measured in this run
                traffic = 2.0 * pj["fetch_bytes_per_launch_raw"] + pj["write_bytes_per_launch_raw"]
                traw = {"fetch_size_raw": pj["fetch_bytes_per_launch_raw"], "write_size": pj["write_bytes_per_launch_raw"]}
                tsrc = (f"{why}; constant from {pmc_rel} (separate --pmc passes over `bench.py --config {args.config} --batch-poses 640`, per "
                        f"launch of this kernel; 2 x FETCH_SIZE + WRITE_SIZE), valid for that workload only")
            else:
                tsrc = f"{why}; no PMC file for this workload"
        # bytes per launch of the form that RUNS (reduce-first: the pair reads the edge records and gathered rows once per kernel and writes the
        # scalar-output message columns once per segment); the fused single-kernel form's bytes stay in the line for comparison
        alg_bytes = (formb.value if formb.value else fb.value) / nl.value
        useful_tf = use.value / (ms.value * 1e-3) / 1e12
        roof = {"bound": "mfma", "achieved": round(ex, 1), "peak": P["peak"], "unit": "TFLOP/s", "frac": round(ex / P["peak"], 4),
                "traffic": traffic, "traffic_source": tsrc, "traffic_counters": traw,
                "algorithmic_bytes_per_launch": alg_bytes, "traffic_ratio": round(traffic / alg_bytes, 3) if traffic else None,
                "what": ("achieved = flops the matrix pipe EXECUTES per second on `instruction`, counted by the library (dbfr_profile_executed_flops); peak = that "
                         "instruction's dense peak; fp32_equivalent_tflops = the REFERENCE algorithm's fp32 flops 2 x 144 x (144 + W) per edge / kernel time -- the "
                         "reduce-first order executes fewer (reference_algorithm_flops_per_executed_flop x 3 products), which is why it may exceed the pipe's "
                         "833-TFLOP/s ceiling for the per-edge order") if mode == "reduce_first" else
                        ("achieved = flops the matrix pipe EXECUTES per second on `instruction` (products_per_fp32_product x the 144 x W share of the "
                         "algorithmic rate); peak = that instruction's dense peak; fp32_equivalent_tflops = algorithmic fp32 flops / kernel time"),
                "kernel": P["kernel"], "instruction": P["instruction"], "products_per_fp32_product": P["products"],
                "executed_tflops_counted": round(exe.value / (ms.value * 1e-3) / 1e12, 1),
                #  what of the executed flops is not padding -- dbfr_profile_useful_flops: the hidden layer once per edge, step A over the
                # edges a segment holds, step B over the segments a unit holds, the (path, u) pairs that exist; x 3 partial products
                "useful_tflops": round(useful_tf, 1), "useful_frac": round(useful_tf / P["peak"], 4),
                "padding_ratio": round(exe.value / use.value, 3) if use.value else None,
                "reference_flops_over_time_tflops": round(alg, 2),
                "fused_form_bytes_per_launch": fb.value / nl.value,
                "reference_algorithm_flops_per_executed_flop": round(fl.value * P["products"] / exe.value, 3) if exe.value else None,
                "fp32_equivalent_tflops": round(alg, 2), "fp32_matrix_peak": FP32_MATRIX_PEAK_TFLOPS,
                "fp32_equivalent_over_fp32_matrix_peak": round(alg / FP32_MATRIX_PEAK_TFLOPS, 4),
                "pipe_sustained_random_operands": P.get("sustained"),
                "frac_of_sustained": round(ex / P["sustained"], 4) if P.get("sustained") else None,
                "sustained_note": ("a bare stream of this instruction with random operands, nothing else running, measured on this board model "
                                   "(tools/exp/mfma_power.hip -> profiles/r3_mfma_power.txt): the power management holds it at 2.05 GHz; the "
                                   "nominal peak is reached with all-zero operands only") if P.get("sustained") else None,
                "launches": nl.value, "avg_launch_ms": round(ms.value / nl.value, 4), "flops_per_launch": fl.value / nl.value,
                "conv_time_share": round(ms.value * 1e-3 / elapsed_s, 4),
                "algorithmic_bytes_note": "the form that runs (dbfr_profile_useful_flops: form_bytes): per edge 4 (48 + 9 + 3 + 48 + 48 + D_in) B of inputs (edge record, "
                                          "two gathered radial-MLP rows, gathered input row) ONCE PER KERNEL of the reduce-first pair, + 4 x the vector-output "
                                          "message columns + 1 flag byte per edge, + 4 x 48 per scalar output irrep per SEGMENT; fused_form_bytes_per_launch = the "
                                          "single-kernel form 4 (48 + 9 + 3 + 48 + 48 + D_in + D_out) per edge.  The [E,W] weights of the reference's two-kernel form "
                                          "are never materialised, so north_star's HBM criterion is superseded by the MFMA bound (DESIGN.md section 5)",
                "algorithmic_gbytes_per_s": round(alg_bytes * nl.value / (ms.value * 1e-3) / 1e9, 1)}
        return roof

    mode = main_mode = model.gemm_mode(dev)
    roof = None if args.no_profile else read_roofline(elapsed, mode)
    if roof is not None and world == 1 and mode == "split_f16" and not args.no_probe:
        # the denominator of frac_of_sustained measured on THIS board: a bare stream of the instruction with random operands for 4 s
        sus = C.c_double()
        if lib.dbfr_probe_mfma_f16(4.0, C.byref(sus), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)) == 0:
            roof["pipe_sustained_random_operands"] = round(sus.value, 1)
            roof["frac_of_sustained"] = round(roof["achieved"] / sus.value, 4)
            roof["sustained_note"] = ("measured in this run on this board (dbfr_probe_mfma_f16: a bare stream of the instruction with random operands on every CU "
                                      "for 4 s, rate over the last 2 s); round 3's board: 1937 (profiles/r3_mfma_power.txt); the nominal peak is reached with "
                                      "all-zero operands only")
    native = None
    if world == 1 and not args.no_profile and not args.no_native and mode != "f32":
        # the same batch through the fp32-matrix-instruction kernels (k_conv / k_conv2), for the record: one batch, untimed warm-up first
        model.set_gemm("f32")
        samp.run_complexes(warm, ppc, dev, seed=0)
        torch.cuda.synchronize(dev)
        lib.dbfr_profile_read(h, None, None, None, None, 1)
        lib.dbfr_profile_enable(h, 1)
        t1 = time.perf_counter()
        samp.run_complexes(warm, ppc, dev, seed=1)
        torch.cuda.synchronize(dev)
        dt1 = time.perf_counter() - t1
        native = read_roofline(dt1, "f32")
        if native:
            native = {"poses_per_sec": round(len(warm) * ppc / dt1, 2), "sample": f"one batch of {len(warm) * ppc} poses, DBFR_GEMM=f32", **native}
        model.set_gemm(mode)
    if rank == 0:
        line = {
            "metric": "poses_per_sec", "value": round(poses / elapsed, 3), "unit": "poses/s",
            "n_gpus": world, "steps": n_steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / n_steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "pose_steps_per_sec": round(poses * T / elapsed, 2),
            "config": {"workload": f"BASELINE.json configs[{args.config - 1}] {cfg['name']}: {len(jobs)} jobs x {ppc} poses "
                                   f"({'one shared ' + cfg['shared'] + ', ' if cfg.get('shared') else ''}~{cfg['n_atoms']} pocket atoms / "
                                   f"~{cfg['n_lig']} ligand atoms, {T} denoise steps per pose) through dist.run_sharded",
                       "batch_poses": B, "batches_this_rank": len(done), "poses_this_rank": int(sum(done)),
                       "batch_seconds_this_rank": [round(b - a, 3) for a, b in zip([t0] + stamps[:-1], stamps)],
                       "poses_total": poses, "denoise_steps": T,
                       "edges_last_step": {"lig": counters[0], "atom": counters[1], "cross": counters[2],
                                           "center": counters[3], "tor": counters[4], "sc_tor": counters[5]},
                       "weights": "seeded random init of the reference architecture (22.9 M params)",
                       "arithmetic": "fp32 in, fp32 out; radial MLP's 144 x W GEMM: " + (PIPE_RF if mode == "reduce_first" else PIPE[mode])["arithmetic"],
                       "parallelism": f"dp{world}: jobs LPT-sharded, poses of a job on one GPU, ragged [ligand | atom14] records gathered at the end "
                                      f"(all_gather_into_tensor in windows of 256 MiB per rank)",
                       "dist_backend": (ddist.dist.get_backend() if ddist.dist.is_initialized() else None),
                       "store": args.store, "gather": args.gather,
                       "edge_budget_regrown_in_timed_region": model.regrown - regrown0,     # DBFR_ERR_CAPACITY -> limits raised -> step resumed
                       "board": board,      # power cap, power and shader clock of rank 0's board during the timed region (amdgpu hwmon files), or None
                       "ranks_seen": len(per_rank),
                       "rank_speed": ([round(v, 4) for v in speeds] if speeds else None),     # relative, from the median of three untimed calibration batches; the LPT shard's weights
                       "per_rank": [{"rank": int(v[0]), "device": int(v[1]), "poses": int(v[2]), "batches": int(v[3]), "elapsed_s": round(v[4], 4),
                                     "sampling_s": round(v[5], 4), "gather_and_unpack_s": round(v[4] - v[5], 4),
                                     "torch_peak_hbm_gib": round(v[6], 3), "records_upload_s_before_timing": round(v[7], 4),
                                     "assemble_host_s": round(v[8], 4),
                                     "poses_per_sec_sampling": round(v[2] / v[5], 2) if v[5] > 0 else None,      # this rank's own rate over its batches
                                     "poses_per_sec_calibration_batch": round(v[9], 2) if v[9] > 0 else None}    # the median of three untimed batches before the timed region
                                    for v in per_rank],
                       # whole-job rate / sum of what the ranks sustained while sampling: what sharding imbalance, the gather and the barriers cost
                       "parallel_efficiency": (round((poses / elapsed) / sum(v[2] / v[5] for v in per_rank if v[5] > 0), 4)
                                               if world > 1 and any(v[5] > 0 for v in per_rank) else None),
                       "hbm_note": "torch_peak_hbm_gib = torch.cuda.max_memory_allocated over the timed region: record halves, packed batch, tapes, "
                                   "trajectories, the library's workspace (a torch tensor); + 0.28 GiB of packed weights the library allocates itself"},
            "roofline": roof,
        }
        if native is not None:
            line["native_f32"] = native
        if args.config == 2 and world == 1:
            # BASELINE.json configs[1] literally (128 complexes x 40 poses = 5120 poses), whatever --steps is: the rate over the first
            # batches that add up to it
            cum = np.cumsum(done)
            k = int(np.searchsorted(cum, 5120))
            if k < len(cum) and cum[k] == 5120:
                line["value_literal_128x40"] = round(5120 / (stamps[k] - t0), 3)
        if world == 1 and not args.no_latency:
            line["latency"] = latency_leg(samp, dev, lib, model)
        if world == 1 and not args.no_cpu_baseline:
            # cfg 1 literally = ONE complex x its 4 poses: (a) one pose through all 20 steps (parity), (b) the 4 poses in one call for 5 steps
            line["cpu_baseline"] = cpu_baseline(args.config, samp, dev, n_poses=args.cpu_poses or 1,
                                                batched={1: (1, 4), 5: None}.get(args.config, (2, 2)), batched_steps=args.cpu_batched_steps)
        print(json.dumps(line), file=line_out, flush=True)
    import torch.distributed as dist
    if dist.is_initialized():      # N > 1 (or the one-rank walk of the collectives, DBFR_DIST_SINGLE=1)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
