#!/usr/bin/env python3
"""Membership-score ranking fidelity of the denoiser precisions (run on the GPU box): per precision, how the ordering of
records by Tartan-Federer score differs from (a) the reference's own scores on the N_tf fixture and (b) the exact-fp32
HIP path on a 100k-record synthetic batch.  Prints one JSON line per (case, precision, timestep)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests.helpers import case_params, load_case, tt  # noqa: E402
from tests.test_gpu_parity import build, ranking_report  # noqa: E402


def main():
    import midst_toolkit_amd as mtk
    from midst_toolkit_amd import scoring

    dev = torch.device("cuda:0")
    precisions = [p for p in ("fp32", "bf16x3", "bf16") if p in mtk.engine.PRECISIONS]
    c = load_case("N_tf")
    x0 = tt(c["x0"]).to(dev)
    noise = tt(c["tf_noise"]).to(dev)
    for prec in precisions:
        d = build(c, precision=prec)
        for k in (0, 1, 3):
            t = int(c[f"tf{k}_t"][0])
            got = scoring.score_records(d, x0, noise, [t], 0, precision=prec)[0]
            want = tt(c[f"tf{k}_score"])
            rep = ranking_report(got, want)
            rep.update(case="N_tf fixture vs reference", precision=prec, t=t, B=int(x0.shape[0]),
                       max_rel_err=float(((got.cpu() - want).abs() / want.abs()).max()))
            print(json.dumps(rep), flush=True)
    # 100k records, 4x512, 20 numeric features (BASELINE configs[1] shape), P = 4
    Nn, T, B, P = 20, 1000, 100_000, 4
    torch.manual_seed(0)
    net = mtk.MLPDiffusion(Nn, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([512] * 4, 0.0))
    diff = mtk.GaussianMultinomialDiffusion(np.array([0]), Nn, net, num_timesteps=T, device=dev).to(dev)
    g = torch.Generator().manual_seed(20250821)
    x = torch.randn((B, Nn), generator=g).clamp(-5.2, 5.2).to(dev)
    nz = torch.randn((P, Nn), generator=g).to(dev)
    ts = [5, 50, 200, 500, 900]
    ref = scoring.score_records(diff, x, nz, ts, 0, precision="fp32")
    for prec in precisions[1:]:
        got = scoring.score_records(diff, x, nz, ts, 0, precision=prec)
        for j, t in enumerate(ts):
            rep = ranking_report(got[j], ref[j])
            rep.update(case="100k records vs fp32 HIP path", precision=prec, t=t, B=B,
                       max_rel_err=float(((got[j] - ref[j]).abs() / ref[j].abs()).max()))
            print(json.dumps(rep), flush=True)


if __name__ == "__main__":
    main()
