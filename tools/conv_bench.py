"""Developer micro-benchmark of ONE fused tensor-product conv through the C ABI test hook (dbfr_test_conv): random
node features / edges of a given count, HIP-event timing, algorithmic TFLOP/s = 2 K (K + W) E / t.
    DBFR_CONV2=1 python tools/conv_bench.py --layer 3 --fam 2 --edges 650000
The knobs of the kernels are environment variables read by the library (DBFR_CONV2, DBFR_GEMM, ...); the timing-only
variants (DBFR_CONV2_BARRIER, DBFR_CONV*_ABL, DBFR_CONV2S_VAR) need a developer build: DBFR_BUILD_DEV=1 python -m diffbindfr_amd.build."""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from diffbindfr_amd import lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--layer", type=int, default=3)
ap.add_argument("--fam", type=int, default=2)
ap.add_argument("--edges", type=int, default=650000)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
model = bench.seeded_params().to(dev)
lib, h = L.load(), model.handle(dev)
W = [2880, 3888, 4896, 7776][min(a.layer, 3)] if a.layer >= 0 else 6912
Din = [48, 84, 120, 168][min(a.layer, 3)] if a.layer >= 0 else 168
Dout = [84, 120, 168, 168][min(a.layer, 3)] if a.layer >= 0 else 96
E = a.edges
N = max(E // 10, 64)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(N, Din, device=dev, generator=g)
xt = torch.randn(N, max(Din, 48), device=dev, generator=g)
tgt = torch.sort(torch.randint(0, N, (E,), device=dev, generator=g)).values.to(torch.int32)
gth = torch.randint(0, N, (E,), device=dev, generator=g).to(torch.int32)
emb = torch.randn(E, 48, device=dev, generator=g)
sh = torch.randn(E, 9, device=dev, generator=g)
ne = torch.tensor([E], dtype=torch.int32, device=dev)
msg = torch.zeros(E, Dout, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def call():
    L.check((lib.dbfr_test_conv2 if os.environ.get('DBFR_CONV2', '0') == '1' else lib.dbfr_test_conv)(h, a.layer, a.fam, E, p(ne), p(tgt), p(gth), p(emb), p(sh), p(xt), xt.shape[1], p(tgt), p(x), Din,
                               p(gth), p(x), Din, p(msg), st))


call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(a.reps):
    e0.record()
    call()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
t = min(ts)
print(f"layer {a.layer} fam {a.fam} E={E} W={W}: best {t:.3f} ms (all {', '.join(f'{v:.2f}' for v in ts)}) = "
      f"{2.0 * 144 * (144 + W) * E / (t * 1e-3) / 1e12:.1f} TFLOP/s  checksum {float(msg.double().abs().sum()):.6e}"
      f"  [CONV2={os.environ.get('DBFR_CONV2', '0')} BARRIER={os.environ.get('DBFR_CONV2_BARRIER', '0')}]")
