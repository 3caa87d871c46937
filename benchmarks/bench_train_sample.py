#!/usr/bin/env python3
"""Secondary measurements: reverse sampling rows/s and training rows/s.

    python benchmarks/bench_train_sample.py [--hidden 512] [--rows 65536] [--batch 4096] [--T 1000]

One JSON line per measurement; shape M (20 numeric + 10 x 10 levels), random-init weights, in-kernel Philox noise.
"""
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
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--rows", type=int, default=65536)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--train-steps", type=int, default=50)
    ap.add_argument("--rng", default="philox", help="philox: in-kernel noise; torch: the reference's draws (the drop-in default)")
    ap.add_argument("--train-only", action="store_true")
    ap.add_argument("--train-precision", default="fp32", choices=["fp32", "bf16"], help="bf16: bf16 operands / fp32 accumulation in the iteration's GEMMs")
    ap.add_argument("--atomics", action="store_true", help="fp32 atomic reductions instead of the deterministic fixed-order ones")
    a = ap.parse_args()
    import midst_toolkit_amd as mtk

    dev = torch.device("cuda:0")
    K, Nn = [10] * 10, 20
    D = Nn + sum(K)
    torch.manual_seed(0)

    def make(precision):
        net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([a.hidden] * 4, 0.0))
        net.precision = precision
        d = mtk.GaussianMultinomialDiffusion(np.asarray(K), Nn, net, num_timesteps=a.T, device=dev).to(dev)
        d.precision = precision
        d.rng = a.rng
        return d

    fl_fwd = 2 * (D * 128 + 128 * a.hidden + 3 * a.hidden * a.hidden + a.hidden * D)
    # ---- reverse sampling: rows / s at T steps (config 4)
    for prec in (() if a.train_only else ("bf16", "fp32")):
        d = make(prec)
        rows = a.rows if prec == "bf16" else max(a.rows // 8, 1024)
        d.sample(1024, torch.ones(1))  # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x, _ = d.sample(rows, torch.ones(1))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"measure": "reverse sampling", "precision": prec, "hidden": a.hidden, "rows": rows, "T": a.T, "seconds": dt,
                          "rows_per_s": rows / dt, "row_steps_per_s": rows * a.T / dt,
                          "denoiser_TFLOPs": rows * a.T * fl_fwd / dt / 1e12}), flush=True)
    # ---- training: rows / s (config 5: fp32 MFMA forward + backward + AdamW + EMA)
    d = make("fp32")
    g = torch.Generator().manual_seed(1)
    x0 = torch.cat([torch.randn((a.batch, Nn), generator=g)] + [torch.randint(0, k, (a.batch, 1), generator=g).float() for k in K], 1).to(dev)

    def batches():
        while True:
            yield x0, None

    tr = mtk.ClavaDDPMTrainer(d, batches(), 2e-4, 1e-4, a.train_steps, device=dev)
    tr.log_every = 10**9
    tr._engine.train_set_deterministic(not a.atomics)
    tr._engine.train_set_precision(a.train_precision)
    for _ in range(6):
        tr._train_step(x0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.train_steps):
        tr._train_step(x0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.train_steps
    fl_step = a.batch * (3 * fl_fwd - 2 * D * 128)
    print(json.dumps({"measure": "training iteration", "precision": a.train_precision, "reductions": "atomics" if a.atomics else "deterministic", "rng": a.rng, "hidden": a.hidden, "batch": a.batch, "ms_per_step": dt * 1e3,
                      "rows_per_s": a.batch / dt, "TFLOPs": fl_step / dt / 1e12, "frac_of_f32_mfma_peak": fl_step / dt / 1e12 / 157.3}), flush=True)


if __name__ == "__main__":
    main()
