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
from .registry import MLDOCK_BUILDER, build_interaction
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
        if scoring_model is not None:
            raise NotImplementedError("the scoring model is outside this path (SURVEY.md section 8 f4)")
        if isinstance(diffusion_model, nn.Module):
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
    def sample_packed(self, pb, noise, visualize=False, sync=True):
        """Run the sampler on an already packed batch.  ``noise``: dict of device tensors
        tr[T,G,3], rot[T,G,3], tor[T,max(NTOR,1)], sc[T,max(NSC,1)].  Returns device tensors
        (lig_traj [T',NL,3], atom14_traj [T',NR,14,3]) with T' = 1 unless ``visualize``."""
        lib = L.load()
        model = self.diffusion_model
        dev = pb.lig_pos.device
        if dev.type != "cuda":
            raise L.DbfrError("DiffBindFRHIP needs a ROCm device (no CPU path)")
        recs, steps = self.schedule()
        T = len(recs)
        d = pb.dims
        ws = model.workspace(pb, dev)
        a14 = torch.zeros(d["NR"], 14, 3, device=dev)
        traj_l = torch.empty(T, d["NL"], 3, device=dev) if visualize else None
        traj_a = torch.empty(T, d["NR"], 14, 3, device=dev) if visualize else None
        nz = L.Noise(*(C.c_void_p(noise[k].data_ptr()) for k in ("tr", "rot", "tor", "sc")))
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        ptr = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
        L.check(lib.dbfr_sample(model.handle(), C.byref(pb.c), steps, T, C.byref(nz), ptr(a14), ptr(traj_l), ptr(traj_a),
                                C.c_void_p(ws.data_ptr()), ws.numel(), C.byref(model.limits), stream))
        if sync:
            L.check(lib.dbfr_status_sync(C.c_void_p(ws.data_ptr()), stream, None))
        if visualize:
            return traj_l, traj_a
        return pb.lig_pos.unsqueeze(0), a14.unsqueeze(0)

    @torch.no_grad()
    def sample_complexes(self, records, poses, device="cuda:0", seed=None, visualize=False, tr_sigma_max=10.0,
                         keep_on_device=False):
        """Records in, poses out: the reference's `_prepare_test_sample` x num_poses + collate + `sample`
        (inference_dataset.py:578-612, struct_init.py, scFlex.py:124-250) with everything per-pose on the device.
        ``records``: list of ``assemble.ComplexRecord`` (or reference-format per-complex dicts); ``poses``: int or
        per-complex list.  Both random tapes (initialisation, SDE noise) come from one torch device generator.
        Returns list[G] of (lig [T,N_l,3], atom14 [T,N_r,14,3]) CPU tensors, complex-major; ``keep_on_device`` leaves them
        in HBM (for ``export.pose_metrics``, which consumes them there)."""
        from . import assemble
        dev = torch.device(device)
        recs = [r if isinstance(r, assemble.ComplexRecord) else assemble.ComplexRecord(r) for r in records]
        pb = assemble.assemble(recs, poses, dev)
        gen = torch.Generator(device=dev)
        if seed is not None:
            gen.manual_seed(int(seed))
        assemble.init_poses(self.diffusion_model, pb, assemble.draw_init_tape(pb, tr_sigma_max, gen))
        steps, _ = self.schedule()
        T, d = len(steps), pb.dims
        z = {"tr": torch.randn(T, pb.G, 3, device=dev, generator=gen), "rot": torch.randn(T, pb.G, 3, device=dev, generator=gen),
             "tor": torch.randn(T, max(d["NTOR"], 1), device=dev, generator=gen),
             "sc": torch.randn(T, max(d["NSC"], 1), device=dev, generator=gen)}
        for s, r in enumerate(steps):
            if r.noise_free:
                for v in z.values():
                    v[s].zero_()
        lig, a14 = self.sample_packed(pb, z, visualize=visualize)
        return self._split(pb, lig, a14) if keep_on_device else self._split(pb, lig.cpu(), a14.cpu())

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
