import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import midst_toolkit_amd as mtk
dev = torch.device("cuda:0")
K, Nn = np.asarray([10] * 10), 20
C = len(K)
D, T, B = Nn + int(K.sum()), 50, 1337
torch.manual_seed(5)
net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([128, 128], 0.0))
d = mtk.GaussianMultinomialDiffusion(K, Nn, net, num_timesteps=T, device=dev).to(dev)
eng = d.engine()
eng.set_math("fast")
g = torch.Generator().manual_seed(9)
out = (torch.randn((D, B), generator=g) * 2.0).to(dev)
z0 = torch.randn((max(Nn, 1), B), generator=g, dtype=torch.float64).to(dev)
zi0 = torch.stack([torch.randint(0, int(k), (B,), generator=g) for k in K]).to(torch.int32).to(dev)
grad = torch.randn((B, Nn), generator=g).to(dev)
noise = torch.randn((B, Nn), generator=g, dtype=torch.float64).to(dev)
def run():
    res = []
    for t, first, gr, nz in ((T - 1, 1, None, None), (17, 0, None, None), (0, 0, None, None), (23, 0, grad, None), (31, 0, None, noise)):
        z, zi, x = z0.clone(), zi0.clone(), torch.zeros((D, B), device=dev)
        eng.p_sample_step(B, t, first, out, (1, B), z, zi, nz, None, x, (1, B), seed=77, row0=4096, grad=gr)
        res.append((z, zi, x))
    return res
new = run()
os.environ["MTK_STEP_ROWS"] = "1"
old = run()
for i, (a, b) in enumerate(zip(new, old)):
    dz = (a[0] != b[0])
    print(i, "z diff", int(dz.sum()), "max rel", float(((a[0] - b[0]).abs() / b[0].abs().clamp_min(1e-30)).max()), "nan", int(torch.isnan(a[0]).sum()), int(torch.isnan(b[0]).sum()),
          "zi diff", int((a[1] != b[1]).sum()), "x diff", int((a[2] != b[2]).sum()))
    if dz.any():
        idx = dz.nonzero()[:5]
        print(idx.tolist(), a[0][dz][:5].tolist(), b[0][dz][:5].tolist())
