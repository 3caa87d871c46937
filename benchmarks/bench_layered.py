"""Layered denoiser alone (shape M, 4 x hidden; --precision bf16 | bf16x3 | fp32; a 4 x 512 bf16 run takes the fused kernel): python benchmarks/bench_layered.py [--rows N] [--hidden H].
A/B builds: MTK_LIB_PATH=<other libmtk_hip.so>; chunking: MTK_LAYERED_CHUNK=<rows>."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import midst_toolkit_amd as mtk  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1 << 20)
    ap.add_argument("--hidden", type=int, default=1024)
    ap.add_argument("--tag", default="")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16x3", "fp32"])
    ap.add_argument("--D", type=int, default=120)
    ap.add_argument("--one-t", action="store_true", help="one timestep for the launch (a reverse-chain step: folded first layer) instead of per-row timesteps")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, D = a.rows, a.D
    torch.manual_seed(0)
    net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([a.hidden] * 4, 0.0)).to(dev)
    net.precision = a.precision
    eng = net.engine(dev)
    x_in = torch.randn((D, B), device=dev)
    out = torch.empty((D, B), device=dev)
    t = 500 if a.one_t else torch.randint(0, 1000, (B,), device=dev)
    fn = lambda: eng.mlp_forward(x_in, t, out=out, precision=a.precision, x_strides=(1, B), out_strides=(1, B), B=B)  # noqa: E731
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for s, e in ev:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    secs = float(np.median([s.elapsed_time(e) for s, e in ev])) * 1e-3
    fl = 2 * (D * 128 + 128 * a.hidden + 3 * a.hidden ** 2 + a.hidden * D)
    print(json.dumps({"tag": a.tag, "lib": os.path.basename(os.environ.get("MTK_LIB_PATH", "libmtk_hip.so")), "chunk": os.environ.get("MTK_LAYERED_CHUNK", "auto"),
                      "one_t": a.one_t, "rows": B, "hidden": a.hidden, "precision": a.precision, "ms": secs * 1e3, "TFLOP/s": B * fl / secs / 1e12, "frac_of_peak": B * fl / secs / (157.3e12 if a.precision == "fp32" else 2.5e15)}), flush=True)


if __name__ == "__main__":
    main()
