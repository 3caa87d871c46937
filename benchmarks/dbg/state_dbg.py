import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import midst_toolkit_amd as mtk
dev = torch.device("cuda:0")
LOGT = float(np.log(np.float32(1e-30)))
def run(K, Nn, layers, B, zero_num=False, all_first=False):
    K = np.asarray(K); C = 0 if int(K[0]) == 0 else len(K); S = int(K.sum()); D = Nn + S
    torch.manual_seed(3)
    net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters(layers, 0.0)); net.precision = "bf16"
    d = mtk.GaussianMultinomialDiffusion(K, Nn, net, num_timesteps=30, device=dev).to(dev); d.precision = "bf16"
    eng = d.engine()
    g = torch.Generator().manual_seed(1)
    z = (torch.randn((max(Nn, 1), B), generator=g, dtype=torch.float64) * 1.7).to(dev)
    if zero_num: z.zero_()
    zi = torch.stack([torch.randint(0, int(k), (B,), generator=g) for k in (K if C else [1])]).to(torch.int32).to(dev)
    if all_first: zi.zero_()
    x = torch.full((D, B), LOGT, dtype=torch.float32, device=dev)
    x[:Nn] = z[:Nn].float()
    off = 0
    for k in range(C):
        x[Nn + off: Nn + off + int(K[k])].scatter_(0, zi[k].long().unsqueeze(0), 0.0); off += int(K[k])
    want = torch.empty((D, B), device=dev); got = torch.full((D, B), float("nan"), device=dev)
    eng.mlp_forward(x, 17, out=want, precision="bf16", x_strides=(1, B), out_strides=(1, B), B=B)
    eng.mlp_forward_state(z, zi, 17, got, B, "bf16", out_strides=(1, B))
    bad = (got != want).any(0)
    print(K.tolist()[:3], Nn, layers[:2], B, "zero_num", zero_num, "all_first", all_first, "rows differing", int(bad.sum()), "max", float((got - want).abs().max()),
          "first bad rows", torch.nonzero(bad)[:8].flatten().tolist(), "bad row tiles", sorted(set((torch.nonzero(bad).flatten() // 256).tolist()))[:40])
for B in (16384, 32768, 65536, 70000):
    run([10] * 10, 20, [1024] * 4, B)
run([10] * 10, 20, [1024] * 4, 32768, zero_num=True)
run([10] * 10, 20, [1024] * 4, 32768, all_first=True)
