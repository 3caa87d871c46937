#!/usr/bin/env python3
"""SURVEY 8(d) cross-check of bench.py's cpu_baseline (in-container only: /root/reference does not travel to the GPU box):
times the REFERENCE's own Tartan-Federer scoring (attacks/tartan_federer/tartan_federer_attack.py: mixed_loss, TF:40-95, over the
reference GaussianMultinomialDiffusion / MLPDiffusion) and the oracle's restatement (oracle.tf_score, what bench.py times as
cpu_baseline "port") on the same records, noise and timesteps, same thread count.  Prints one JSON line; the two rates should
agree within +-20 % - the oracle is then a fair stand-in for the reference on the GPU box's host cores."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ref_shim  # noqa: E402
from oracle import diffusion_oracle as orc  # noqa: E402


def main():
    TF = ref_shim.load_tf()
    gmd, _, model = ref_shim.load()
    from midst_toolkit.common.enumerations import IsTargetConditioned

    Nn, T, n_rec, P = 20, 1000, 8192, 1
    params = orc.init_mlp_params(Nn, [512] * 4, seed=0)
    net = model.MLPDiffusion(Nn, 0, IsTargetConditioned.NONE, model.DiffusionParameters(layers_dimensions=[512] * 4, dropout=0.0))
    net.load_state_dict(params, strict=True)
    net.eval()
    diff = gmd.GaussianMultinomialDiffusion(np.asarray([0]), Nn, net, num_timesteps=T, scheduler_type=gmd.SchedulerType.COSINE,
                                            device=torch.device("cpu"))
    diff.eval()
    s = orc.make_schedule(T, "cosine")
    g = torch.Generator().manual_seed(0)
    x = torch.randn((n_rec, Nn), generator=g).clamp(-5.2, 5.2)
    noise = torch.randn((P, Nn), generator=g)
    ts = [(37 * i + 1) % T for i in range(24)]
    y = {"y": torch.zeros(n_rec * P, dtype=torch.long)}
    res = {}
    fns = {"reference": lambda t: TF.mixed_loss(diff, x, y, noise.tolist(), P, 0, t), "oracle": lambda t: orc.tf_score(s, Nn, params, x, noise, t)}
    spent = {k: 0.0 for k in fns}
    with torch.no_grad():
        for fn in fns.values():
            fn(3)
        for rep in range(3):  # interleaved, in both orders: whichever runs first after a pause pays for cold caches and idle threads
            for name in (("reference", "oracle") if rep % 2 == 0 else ("oracle", "reference")):
                t0 = time.perf_counter()
                outs = [fns[name](t) for t in ts]
                spent[name] += time.perf_counter() - t0
                res[name + "_last"] = outs[-1]
    for name in fns:
        res[name] = 3 * n_rec * len(ts) / spent[name]
    a, b = res.pop("reference_last"), res.pop("oracle_last")
    same = bool(torch.equal(torch.as_tensor(a).reshape(-1).float(), torch.as_tensor(b).reshape(-1).float()))
    print(json.dumps({"reference_evals_per_s": res["reference"], "oracle_evals_per_s": res["oracle"], "ratio": res["oracle"] / res["reference"],
                      "threads": torch.get_num_threads(), "records": n_rec, "timesteps": len(ts), "scores_bit_identical": same}))


if __name__ == "__main__":
    main()
