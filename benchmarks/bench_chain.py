#!/usr/bin/env python3
"""Reverse chain at shape M: rows/s with the denoiser reading the chain's state (mtk_mlp_forward_state) against the materialised
input matrix, same seed.  python benchmarks/bench_chain.py [--hidden 1024] [--rows 262144] [--T 200]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=1024)
    ap.add_argument("--rows", type=int, default=262144)
    ap.add_argument("--T", type=int, default=200)
    ap.add_argument("--routes", default="state,matrix")
    a = ap.parse_args()
    import midst_toolkit_amd as mtk

    dev = torch.device("cuda:0")
    K, Nn = np.array([10] * 10), 20
    D = Nn + int(K.sum())
    torch.manual_seed(0)
    net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([a.hidden] * 4, 0.0))
    net.precision = "bf16"
    d = mtk.GaussianMultinomialDiffusion(K, Nn, net, num_timesteps=a.T, device=dev).to(dev)
    d.precision, d.rng, d.results_on_device = "bf16", "philox", True
    fl = 2 * (D * 128 + 128 * a.hidden + 3 * a.hidden * a.hidden + a.hidden * D)
    for route in a.routes.split(","):
        d.state_input = route == "state"
        d.sample(a.rows, torch.ones(1), seed=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x, _ = d.sample(a.rows, torch.ones(1), seed=2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"route": route, "hidden": a.hidden, "rows": a.rows, "T": a.T, "ms_per_step": dt / a.T * 1e3,
                          "rows_per_s_at_T1000": a.rows / (dt / a.T * 1000), "mfma_frac": a.rows * a.T * fl / dt / 2.5e15,
                          "checksum": float(x.double().sum())}), flush=True)


if __name__ == "__main__":
    main()
