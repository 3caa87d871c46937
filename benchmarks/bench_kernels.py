#!/usr/bin/env python3
"""Per-kernel roofline numbers for the HBM-bound diffusion kernels and the mixed (shape M) paths.

python benchmarks/bench_kernels.py [--rows 1048576] [--hidden 512]
Prints one JSON line per measurement: algorithmic bytes / HIP-event time.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import midst_toolkit_amd as mtk  # noqa: E402

HBM_PEAK = 8000.0  # GB/s


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in ev:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    return float(np.median([s.elapsed_time(e) for s, e in ev])) * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1 << 20)
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--levels", type=int, default=10, help="levels per categorical block (SURVEY section 8's stress variant: 100 -> S = 1000, D = 1020)")
    ap.add_argument("--hbm-only", action="store_true", help="only the three HBM-bound diffusion kernels, fast math (PMC passes)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, Nn, K, T = a.rows, 20, np.array([a.levels] * 10), 1000
    C, S = len(K), int(K.sum())
    D = Nn + S
    torch.manual_seed(0)
    net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([a.hidden] * 4, 0.0))
    d = mtk.GaussianMultinomialDiffusion(K, Nn, net, num_timesteps=T, device=dev).to(dev)
    d.rng = "philox"  # in-kernel noise => MTK_MATH_FAST
    eng = d.engine()
    g = torch.Generator().manual_seed(20250821)
    x0 = torch.cat([torch.randn((B, Nn), generator=g).clamp(-5.2, 5.2)] + [torch.randint(0, int(k), (B, 1), generator=g).float() for k in K], 1).to(dev)
    t = torch.randint(0, T, (B,), generator=g).to(dev)
    x_in = torch.empty((D, B), device=dev)
    out = torch.randn((D, B), device=dev)
    idx = torch.empty((C, B), dtype=torch.int32, device=dev)
    lm, lg = torch.empty(B, device=dev), torch.empty(B, device=dev)

    def emit(name, secs, bytes_per_row, note=""):
        gbs = B * bytes_per_row / secs / 1e9
        print(json.dumps({"kernel": name, "rows": B, "ms": secs * 1e3, "bytes_per_row": bytes_per_row, "GB/s": gbs,
                          "frac_of_8TBs": gbs / HBM_PEAK, "rows_per_s": B / secs, "note": note}), flush=True)

    for math in (("fast",) if a.hbm_only else ("fast", "reference")):
      eng.set_math(math)
      tag = f"Philox, math={math}"
      # q_sample, Philox noise: read x0 (Nn + C floats) + t; write model input D floats + C idx
      b_q = (Nn + C) * 4 + 8 + D * 4 + C * 4
      emit(f"mtk_q_sample ({tag})", timeit(lambda: eng.q_sample(x0, t, None, None, x_in, (1, B), idx, seed=1)), b_q)
      # loss_rows, Philox eps: read out D floats + x0 + idx + t; write 2 floats
      b_l = D * 4 + (Nn + C) * 4 + C * 4 + 8 + 8
      emit(f"mtk_loss_rows ({tag})", timeit(lambda: eng.loss_rows(out, (1, B), x0, idx, None, t, lm, lg, seed=1)), b_l)
      # reverse step: z fp64 rw + eps_hat + out_cat + idx rw + next model input
      z = torch.randn((Nn, B), device=dev, dtype=torch.float64)
      zi = torch.randint(0, a.levels, (C, B), device=dev, dtype=torch.int32)
      b_s = Nn * 8 * 2 + Nn * 4 + S * 4 + C * 4 * 2 + D * 4
      emit(f"mtk_p_sample_step ({tag})", timeit(lambda: eng.p_sample_step(B, 500, False, out, (1, B), z, zi, None, None, x_in, (1, B), seed=1)), b_s)
    eng.set_math("fast")
    if a.hbm_only:
        return
    # denoiser alone, shape M
    for prec in ("bf16", "fp32"):
        secs = timeit(lambda: eng.mlp_forward(x_in, t, out=out, precision=prec, x_strides=(1, B), out_strides=(1, B), B=B), iters=5)
        fl = 2 * (D * 128 + 128 * a.hidden + 3 * a.hidden ** 2 + a.hidden * D)
        peak = 2500.0 if prec == "bf16" else 157.3
        print(json.dumps({"kernel": f"mtk_mlp_forward {prec} (shape M, 4x{a.hidden})", "rows": B, "ms": secs * 1e3, "TFLOP/s": B * fl / secs / 1e12,
                          "frac_of_peak": B * fl / secs / 1e12 / peak, "rows_per_s": B / secs}), flush=True)
    # whole mixed_loss (Philox) and reverse row-steps per second
    for prec in ("bf16", "fp32"):
        d.precision = prec
        secs = timeit(lambda: d.mixed_loss_rows(x0, timestep=t, seed=7), iters=5)
        print(json.dumps({"path": f"mixed_loss_rows {prec} (q_sample + denoiser + loss_rows)", "rows": B, "ms": secs * 1e3, "evals_per_s": B / secs}), flush=True)

        def rev():
            eng.mlp_forward(x_in, 500, out=out, precision=prec, x_strides=(1, B), out_strides=(1, B), B=B)
            eng.p_sample_step(B, 500, False, out, (1, B), z, zi, None, None, x_in, (1, B), seed=1)

        secs = timeit(rev, iters=5)
        print(json.dumps({"path": f"reverse step {prec} (denoiser + p_sample_step)", "rows": B, "ms": secs * 1e3, "row_steps_per_s": B / secs,
                          "rows_per_s_at_T1000": B / secs / 1000}), flush=True)


if __name__ == "__main__":
    main()
