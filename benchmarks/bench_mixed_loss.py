"""mixed_loss_rows end to end (q_sample + fused denoiser + loss_rows), shape M, 4x512, in-kernel noise."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midst_toolkit_amd as mtk  # noqa: E402

dev = torch.device("cuda:0")
B, Nn, K, T, H = 1 << 20, 20, np.array([10] * 10), 1000, int(sys.argv[1]) if len(sys.argv) > 1 else 512
D = Nn + int(K.sum())
torch.manual_seed(0)
net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([H] * 4, 0.0))
d = mtk.GaussianMultinomialDiffusion(K, Nn, net, num_timesteps=T, device=dev).to(dev)
d.rng, d.precision = "philox", "bf16"
net.precision = "bf16"
g = torch.Generator().manual_seed(1)
x0 = torch.cat([torch.randn((B, Nn), generator=g)] + [torch.randint(0, int(k), (B, 1), generator=g).float() for k in K], 1).to(dev)
t = torch.randint(0, T, (B,), generator=g).to(dev)
for _ in range(3):
    d.mixed_loss_rows(x0, timestep=t, seed=7)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    lm, lg = d.mixed_loss_rows(x0, timestep=t, seed=7)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(json.dumps({"path": "mixed_loss_rows bf16", "hidden": H, "rows": B, "ms": dt * 1e3, "evals_per_s": B / dt}))
