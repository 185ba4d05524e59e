"""``DiffBindFRHIP``: drop-in for MLDOCK_BUILDER['DiffBindFR'] (level-2 boundary).

Same constructor and ``forward(data, mode='test', visualize=False)`` contract as
druglib/models/Docking/scFlex.py:26-250 / druglib/models/Docking/base.py:139-151; the
whole 20-step reverse SDE (score network, Euler-Maruyama perturbations, ligand
rigid+torsion update with Kabsch re-alignment, chi update and side-chain rebuild) runs
on the device inside one ``dbfr_sample`` call -- no per-step Python, no deepcopy, no
host<->device traffic besides the pre-drawn noise tape.
"""
import ctypes as C

import torch
from torch import nn

from . import lib as L
from . import schedule
from .packing import PackedBatch, _get, _has
from .registry import MLDOCK_BUILDER, build_energy, build_interaction
from .score_model import TensorProductModelHIP, cfg_get


def draw_noise_tape(recs, G, n_tor, n_sc, generator=None):
    """N(0,1) tape drawn on the CPU in the reference's order: per step tr, rot, tor,
    sc_tor (scFlex.py:167-183,202-205); nothing is drawn on noise-free steps."""
    T = len(recs)
    z = dict(tr=torch.zeros(T, G, 3), rot=torch.zeros(T, G, 3), tor=torch.zeros(T, max(n_tor, 1)),
             sc=torch.zeros(T, max(n_sc, 1)))
    for s, r in enumerate(recs):
        if r.noise_free:
            continue
        z["tr"][s] = torch.normal(mean=0, std=1, size=(G, 3), generator=generator)
        z["rot"][s] = torch.normal(mean=0, std=1, size=(G, 3), generator=generator)
        if n_tor:
            z["tor"][s] = torch.normal(mean=0, std=1, size=(n_tor,), generator=generator)
        if n_sc:
            z["sc"][s] = torch.normal(mean=0, std=1, size=(n_sc,), generator=generator)
    return z


@MLDOCK_BUILDER.register_module(name=["DiffBindFRHIP"])
class DiffBindFRHIP(nn.Module):
    def __init__(self, diffusion_model=None, scoring_model=None, train_cfg=None, test_cfg=None, pretrained=None,
                 init_cfg=None, **kwargs):
        super().__init__()
        if scoring_model is not None:          # scFlex.py:43-46: built through the ENERGY registry (``mdn.KarmaDockHIP`` is registered there)
            if isinstance(scoring_model, nn.Module):
                self.scoring_model = scoring_model
            else:
                from . import mdn  # noqa: F401  (registers KarmaDockHIP)
                sm = dict(scoring_model)
                if sm.get("type") in (None, "KarmaDock"):
                    sm["type"] = "KarmaDockHIP"
                self.scoring_model_cfg = sm.get("cfg")
                self.scoring_model = build_energy(sm)
        if diffusion_model is None:
            pass                                   # scFlex.py:39: a scorer-only model is legal
        elif isinstance(diffusion_model, nn.Module):
            self.diffusion_model = diffusion_model
        else:
            dm = dict(diffusion_model)
            if dm.get("type") in (None, "TensorProductModel"):
                dm["type"] = "TensorProductModelHIP"
            self.diffusion_model_cfg = dm.get("cfg")
            self.diffusion_model = build_interaction(dm)
        self.train_cfg, self.test_cfg = train_cfg or {}, test_cfg or {}
        self.torus_seed = 0
        self._sched = None

    # base.py:139-151
    def forward(self, data, mode="test", **kwargs):
        if mode != "test":
            raise NotImplementedError("DiffBindFRHIP is an inference drop-in (mode='test')")
        return self.forward_test(data, **kwargs)

    def forward_test(self, data, *args, **kwargs):
        if hasattr(data, "to_dict"):
            data = data.to_dict(decode=True, drop_meta=False)   # scFlex.py:72-75
        return self.sample(data, *args, **kwargs)

    def sample_cfg(self):
        return schedule.sample_cfg(cfg_get(self.test_cfg, "sample_cfg", None))

    def schedule(self):
        if self._sched is None:
            self._sched = schedule.steps(self.sample_cfg(), self.torus_seed)
        return self._sched

    @torch.no_grad()
    def sample_packed(self, pb, noise, visualize=False, sync=True, stop=None):
        """Run the sampler on an already packed batch.  ``noise``: dict of device tensors
        tr[T,G,3], rot[T,G,3], tor[T,max(NTOR,1)], sc[T,max(NSC,1)].  Returns device tensors
        (lig_traj [T',NL,3], atom14_traj [T',NR,14,3]) with T' = 1 unless ``visualize``.  ``stop``: run steps [0, stop) only
        (dbfr_sample_range); ``pb`` then holds the state entering step ``stop``."""
        lib = L.load()
        model = self.diffusion_model
        dev = pb.lig_pos.device
        if dev.type != "cuda":
            raise L.DbfrError("DiffBindFRHIP needs a ROCm device (no CPU path)")
        recs, steps = self.schedule()
        T = len(recs) if stop is None else max(1, min(int(stop), len(recs)))
        d = pb.dims
        with torch.cuda.device(dev):        # everything the library creates (streams, events, weights) follows the current device
            a14 = torch.zeros(d["NR"], 14, 3, device=dev)
            traj_l = torch.empty(T, d["NL"], 3, device=dev) if visualize else None
            traj_a = torch.empty(T, d["NR"], 14, 3, device=dev) if visualize else None
            nz = L.Noise(*(C.c_void_p(noise[k].data_ptr()) for k in ("tr", "rot", "tor", "sc")))
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            ptr = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
            first = 0
            while True:
                ws = model.workspace(pb, dev)
                L.check(lib.dbfr_sample_range(model.handle(dev), C.byref(pb.c), steps, T, first, C.byref(nz), ptr(a14), ptr(traj_l),
                                              ptr(traj_a), C.c_void_p(ws.data_ptr()), ws.numel(), C.byref(model.limits), stream))
                if not sync:
                    break
                rc = lib.dbfr_status_sync(C.c_void_p(ws.data_ptr()), stream, None)
                if rc == L.DBFR_ERR_CAPACITY and model.auto_grow:
                    # an edge list of step `first` outgrew its budget: the device froze the poses at the beginning of that
                    # step; re-plan the workspace with the counted sizes and resume there (no step is computed twice)
                    first = model.grow_limits(pb, ws, stream)
                    continue
                L.check(rc)
                break
        if visualize:
            return traj_l, traj_a
        return pb.lig_pos.unsqueeze(0), a14.unsqueeze(0)

    # ---- random tapes.  One generator stream PER JOB (complex), seeded from (seed, job id): the poses of a job do not
    # depend on which other jobs share its batch or on which rank runs it, so a sharded run reproduces the 1-GPU run.
    @staticmethod
    def job_seed(seed, job_id, chunk=0):
        x = (int(seed) * 0x9E3779B97F4A7C15 + int(job_id) * 0xBF58476D1CE4E5B9 + int(chunk) * 0x94D049BB133111EB + 1) & (2 ** 64 - 1)
        x ^= x >> 30; x = (x * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
        x ^= x >> 27; x = (x * 0x94D049BB133111EB) & (2 ** 64 - 1)
        x ^= x >> 31
        return x & (2 ** 63 - 1)

    def draw_tapes(self, records, poses, seeds, dev, tr_sigma_max=10.0, pose_ranges=None, tapes=None):
        """Initialisation tape (``assemble.draw_init_tape_dims`` layout) and SDE noise tape of a complex-major batch, drawn
        complex by complex on the device from ``torch.Generator(seed_c)``.  ``pose_ranges[c] = (p0, P_total)``: the batch
        holds poses ``p0 .. p0 + poses[c]`` of a job of ``P_total`` poses -- the job's WHOLE tape is drawn and the rows of
        these poses are cut out, so a job's poses do not depend on how the driver cut it into batches.
        ``tapes[c] = (init, z)`` (or None): RECORDED tapes of job c used instead of drawing -- ``init`` as
        ``assemble.draw_init_tape_dims(P_total, ...)`` returns it (tor [P n_tor], rot [P,3,3], tr [P,3], sc [P n_r,4]), ``z`` =
        dict(tr [T,P,3], rot [T,P,3], tor [T,P n_tor], sc [T,P n_sc]); replays a run (tests: the reference's own trajectories)."""
        from . import assemble
        recs_, _ = self.schedule()
        T = len(recs_)
        reps = [poses] * len(records) if isinstance(poses, int) else list(poses)
        init = {k: [] for k in ("tor", "rot", "tr", "sc")}
        z = {k: [] for k in ("tr", "rot", "tor", "sc")}
        gen = torch.Generator(device=dev)
        for c, (r, P, s) in enumerate(zip(records, reps, seeds)):
            p0, Pt = (0, P) if pose_ranges is None else pose_ranges[c]
            assert 0 <= p0 and p0 + P <= Pt
            if tapes is not None and tapes[c] is not None:
                f = lambda x, *shape: torch.as_tensor(x).to(device=dev, dtype=torch.float32).reshape(*shape)
                t = {k: f(tapes[c][0][k], *shp) for k, shp in (("tor", (Pt * r.n_tor,)), ("rot", (Pt, 3, 3)), ("tr", (Pt, 3)), ("sc", (Pt * r.n_r, 4)))}
                zz = {k: f(tapes[c][1][k], T, *shp) for k, shp in (("tr", (Pt, 3)), ("rot", (Pt, 3)), ("tor", (Pt * r.n_tor,)), ("sc", (Pt * r.n_sc,)))}
            else:
                gen.manual_seed(int(s))
                t = assemble.draw_init_tape_dims(Pt, Pt * r.n_tor, Pt * r.n_r, dev, tr_sigma_max, gen)
                zz = dict(tr=torch.randn(T, Pt, 3, device=dev, generator=gen), rot=torch.randn(T, Pt, 3, device=dev, generator=gen),
                          tor=torch.randn(T, Pt * r.n_tor, device=dev, generator=gen), sc=torch.randn(T, Pt * r.n_sc, device=dev, generator=gen))
            init["tor"].append(t["tor"][p0 * r.n_tor:(p0 + P) * r.n_tor])
            init["rot"].append(t["rot"][p0:p0 + P])
            init["tr"].append(t["tr"][p0:p0 + P])
            init["sc"].append(t["sc"][p0 * r.n_r:(p0 + P) * r.n_r])
            z["tr"].append(zz["tr"][:, p0:p0 + P])
            z["rot"].append(zz["rot"][:, p0:p0 + P])
            z["tor"].append(zz["tor"][:, p0 * r.n_tor:(p0 + P) * r.n_tor])
            z["sc"].append(zz["sc"][:, p0 * r.n_sc:(p0 + P) * r.n_sc])
        init = {k: torch.cat(v, 0) for k, v in init.items()}
        z = {k: torch.cat(v, 1) for k, v in z.items()}
        for k in ("tor", "sc"):
            if z[k].shape[1] == 0:
                z[k] = torch.zeros(T, 1, device=dev)
        if init["tor"].numel() == 0:
            init["tor"] = torch.zeros(1, device=dev)
        for s_, r_ in enumerate(recs_):
            if r_.noise_free:
                for v in z.values():
                    v[s_].zero_()
        return {k: v.contiguous() for k, v in init.items()}, {k: v.contiguous() for k, v in z.items()}

    @torch.no_grad()
    def sample_complexes(self, records, poses, device="cuda:0", seed=None, visualize=False, tr_sigma_max=10.0,
                         keep_on_device=False, job_ids=None, seeds=None):
        """Records in, poses out: the reference's `_prepare_test_sample` x num_poses + collate + `sample`
        (inference_dataset.py:578-612, struct_init.py, scFlex.py:124-250) with everything per-pose on the device.
        ``records``: list of ``assemble.ComplexRecord`` (or reference-format per-complex dicts); ``poses``: int or
        per-complex list.  Both random tapes (initialisation, SDE noise) come from per-complex device generators seeded
        from (``seed``, ``job_ids[c]``) (or given outright as ``seeds``); ``seed=None`` takes a fresh seed from torch's global generator (the reference
        draws from the global RNG), so two unseeded calls give different poses.
        Returns list[G] of (lig [T,N_l,3], atom14 [T,N_r,14,3]) CPU tensors, complex-major; ``keep_on_device`` leaves them
        in HBM (for ``export.pose_metrics``, which consumes them there)."""
        pb, lig, a14 = self.run_complexes(records, poses, device, seed, visualize, tr_sigma_max, job_ids, seeds)
        return self._split(pb, lig, a14) if keep_on_device else self._split(pb, lig.cpu(), a14.cpu())

    @torch.no_grad()
    def run_complexes(self, records, poses, device="cuda:0", seed=None, visualize=False, tr_sigma_max=10.0, job_ids=None,
                      seeds=None, pose_ranges=None, tapes=None, stop=None):
        """``sample_complexes`` without the per-graph split: (packed batch, lig [T,NL,3], atom14 [T,NR,14,3]) on the device;
        graphs are complex-major, so the poses of complex c are rows ``pb.lig_ptr_host[g0] .. [g0 + poses_c]``.
        ``pose_ranges`` / ``tapes``: see ``draw_tapes`` (a job cut into several batches by ``dist.run_sharded``; recorded tapes)."""
        from . import assemble
        dev = torch.device(device)
        recs = [r if isinstance(r, assemble.ComplexRecord) else assemble.ComplexRecord(r) for r in records]
        if seeds is None:
            if seed is None:
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            job_ids = list(range(len(recs))) if job_ids is None else list(job_ids)
            seeds = [self.job_seed(seed, j) for j in job_ids]
        with torch.cuda.device(dev):
            pb = assemble.assemble(recs, poses, dev)
            init, z = self.draw_tapes(recs, poses, seeds, dev, tr_sigma_max, pose_ranges, tapes)
            assemble.init_poses(self.diffusion_model, pb, init)
            lig, a14 = self.sample_packed(pb, z, visualize=visualize, stop=stop)
        return pb, lig, a14

    def _split(self, pb, lig, a14):
        lp, rp = pb.lig_ptr_host.tolist(), pb.res_ptr_host.tolist()
        out = [(lig[:, lp[g]:lp[g + 1]].clone(), a14[:, rp[g]:rp[g + 1]].clone()) for g in range(pb.G)]
        if cfg_get(self.diffusion_model_cfg if hasattr(self, "diffusion_model_cfg") else None, "no_sc_torsion", False):
            return [o[0] for o in out]
        return out

    @torch.no_grad()
    def sample(self, data, visualize=False):
        """scFlex.py:124-250.  Returns list[G] of (lig [T,N_l,3], atom14 [T,N_r,14,3]) CPU tensors."""
        dev = TensorProductModelHIP._device_of(data)
        pb = PackedBatch(data, dev)
        recs, _ = self.schedule()
        z = draw_noise_tape(recs, pb.G, pb.dims["NTOR"], pb.dims["NSC"])
        z = {k: v.to(dev).contiguous() for k, v in z.items()}
        lig, a14 = self.sample_packed(pb, z, visualize=visualize)
        return self._split(pb, lig.cpu(), a14.cpu())
