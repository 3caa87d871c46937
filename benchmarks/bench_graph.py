#!/usr/bin/env python3
"""Sampled rows/s of the reverse chain at small batch sizes, eager T-loop (2 T + 2 launches from Python) vs the same
chain replayed as one captured HIP graph (`use_graph`): python benchmarks/bench_graph.py [--hidden 512 | --layers 512,1024,1024,1024,1024,512]
[--T 1000].  The --layers example is ClavaDDPM's default denoiser.  MTK_* switches of DESIGN section 5 select the general kernels."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import midst_toolkit_amd as mtk  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--layers", default="", help="comma-separated hidden widths (overrides --hidden)")
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--rows", default="256,2048,8192,32768")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    K, Nn = np.array([10] * 10), 20
    D = Nn + int(K.sum())
    torch.manual_seed(0)
    layers = [int(v) for v in a.layers.split(",")] if a.layers else [a.hidden] * 4
    name = "x".join(str(v) for v in layers) if a.layers else f"4x{a.hidden}"
    net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters(layers, 0.0))
    net.precision = "bf16"
    d = mtk.GaussianMultinomialDiffusion(K, Nn, net, num_timesteps=a.T, device=dev).to(dev)
    d.precision, d.rng, d.results_on_device = "bf16", "philox", True
    for B in [int(v) for v in a.rows.split(",")]:
        res = {}
        for mode in ("eager", "graph"):
            d.use_graph = mode == "graph"
            d.sample(B, torch.ones(1), seed=1)  # warm-up (graph: capture)
            torch.cuda.synchronize()
            ts = []
            for rep in range(3):
                t0 = time.perf_counter()
                d.sample(B, torch.ones(1), seed=2 + rep)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            res[mode] = min(ts)
        print(json.dumps({"rows": B, "T": a.T, "denoiser": f"{name} bf16", "eager_s": res["eager"], "graph_s": res["graph"],
                          "eager_rows_per_s": B / res["eager"], "graph_rows_per_s": B / res["graph"], "speedup": res["eager"] / res["graph"],
                          "us_per_step_eager": res["eager"] / a.T * 1e6, "us_per_step_graph": res["graph"] / a.T * 1e6}), flush=True)


if __name__ == "__main__":
    main()
