"""Tartan-Federer scoring throughput with ClavaDDPM's default denoiser (512, 1024 x 4, 512: layered path) beside 4 x 512 (fused kernel)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import midst_toolkit_amd as mtk
from midst_toolkit_amd import scoring

dev = torch.device("cuda:0")
Nn, N, nt = 20, 100000, 50
for name, layers in (("4x512", [512] * 4), ("clavaddpm_default", [512, 1024, 1024, 1024, 1024, 512])):
    for prec in ("bf16", "bf16x3"):
        torch.manual_seed(0)
        net = mtk.MLPDiffusion(Nn, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters(list(layers), 0.0))
        net.precision = prec
        d = mtk.GaussianMultinomialDiffusion(np.asarray([0]), Nn, net, num_timesteps=1000, device=dev).to(dev)
        d.precision = prec
        x = torch.randn((N, Nn), device=dev)
        noise = torch.randn((1, Nn))
        ts = list(range(0, 1000, 1000 // nt))
        scoring.score_records(d, x, noise, ts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            s = scoring.score_records(d, x, noise, ts)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        fl = 2 * sum(a * b for a, b in zip([Nn, 128] + layers, [128] + layers + [Nn]))
        print(json.dumps({"denoiser": name, "precision": prec, "evals_per_s": N * len(ts) / dt, "TFLOP/s": N * len(ts) * fl / dt / 1e12, "finite": bool(torch.isfinite(s).all())}), flush=True)
