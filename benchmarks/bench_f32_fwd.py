"""Exact-fp32 denoiser forward, shape M: python benchmarks/bench_f32_fwd.py [hidden]  (A/B: MTK_LIB_PATH)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midst_toolkit_amd as mtk  # noqa: E402

dev = torch.device("cuda:0")
H = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B, D = 1 << 19, 120
torch.manual_seed(0)
net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([H] * 4, 0.0)).to(dev)
eng = net.engine(dev)
x_in, out = torch.randn((D, B), device=dev), torch.empty((D, B), device=dev)
t = torch.randint(0, 1000, (B,), device=dev)
fn = lambda: eng.mlp_forward(x_in, t, out=out, precision="fp32", x_strides=(1, B), out_strides=(1, B), B=B)  # noqa: E731
for _ in range(2):
    fn()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(6)]
for s, e in ev:
    s.record(); fn(); e.record()
torch.cuda.synchronize()
secs = float(np.median([s.elapsed_time(e) for s, e in ev])) * 1e-3
fl = 2 * (D * 128 + 128 * H + 3 * H * H + H * D)
print(json.dumps({"lib": os.path.basename(os.environ.get("MTK_LIB_PATH", "libmtk_hip.so")), "hidden": H, "rows": B, "ms": secs * 1e3,
                  "TFLOP/s": B * fl / secs / 1e12, "frac_of_157": B * fl / secs / 157.3e12}))
