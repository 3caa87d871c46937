"""The oracle's training restatement (loss_and_grads / adamw_update / train_loop) against the reference's
ClavaDDPMTrainer run captured in tests/golden/train_*.npz (tools/make_golden.py, gen_train)."""
import os

import numpy as np
import pytest
import torch

from oracle import diffusion_oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_train(tag):
    z = np.load(os.path.join(GOLD, f"train_{tag}.npz"))
    c = {k: z[k] for k in z.files}
    c["K"] = [int(k) for k in c["K"]]
    c["Nn"], c["T"], c["steps"] = int(c["Nn"]), int(c["T"]), int(c["steps"])
    c["d_layers"] = [int(v) for v in c["d_layers"]]
    c["D"] = c["Nn"] + sum(c["K"])
    c["batches"] = [(torch.from_numpy(c["x"][i]), torch.from_numpy(c["t"][i]), torch.from_numpy(c["eps"][i]), torch.from_numpy(c["u"][i]))
                    for i in range(c["steps"])]
    return c


def flatten(params, n_layers):
    return torch.cat([torch.cat([params[n + ".weight"].reshape(-1), params[n + ".bias"].reshape(-1)]) for n in orc.mlp_param_names(n_layers)])


@pytest.mark.parametrize("tag", ["B_ragged", "A_numeric"])
def test_train_loop_matches_reference(tag):
    c = load_train(tag)
    s = orc.make_schedule(c["T"], "cosine")
    p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
    hist, p, ema = orc.train_loop(s, c["K"], c["Nn"], p0, c["batches"], float(c["lr"]), float(c["wd"]), c["steps"])
    np.testing.assert_array_equal(hist.astype(np.float32), c["losses"].astype(np.float32))
    n = len(c["d_layers"])
    np.testing.assert_array_equal(flatten(p, n).numpy(), c["final"])
    np.testing.assert_array_equal(flatten(ema, n).numpy(), c["ema"])


def test_adamw_update_is_torch_adamw():
    g = torch.Generator().manual_seed(3)
    p = torch.randn((37, 11), generator=g)
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([ref], lr=2e-3, weight_decay=1e-4)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 6):
        gr = torch.randn((37, 11), generator=g)
        ref.grad = gr.clone()
        opt.step()
        orc.adamw_update(p, gr, m, v, step, 2e-3, 1e-4)
        assert torch.equal(p, ref.detach())


def load_berka():
    z = np.load(os.path.join(GOLD, "train_berka.npz"))
    c = {k: z[k] for k in z.files}
    c["K"], c["Nn"], c["T"], c["steps"], c["stride"] = [0], int(c["Nn"]), int(c["T"]), int(c["steps"]), int(c["stride"])
    c["d_layers"] = [int(v) for v in c["d_layers"]]
    c["D"] = c["Nn"]
    table = torch.from_numpy(c["table"])
    none = torch.zeros((c["order"].shape[1], 0))
    c["batches"] = [(table[torch.from_numpy(c["order"][i])], torch.from_numpy(c["t"][i]), torch.from_numpy(c["eps"][i]), none) for i in range(c["steps"])]
    return c


def test_train_loop_matches_reference_on_the_berka_run():
    """BASELINE configs[0] / SURVEY 8c: 200 iterations of the reference trainer on the Berka 'trans' subset with the
    integration test's denoiser [512, 1024, 1024, 1024, 1024, 512] (tests/golden/train_berka.npz): the oracle's loss curve
    and a strided sample of the trained / EMA weights.  Bit-exact on the CPU that generated the fixture; on another host
    its fp32 BLAS roundings differ, so the bar here is the free-running one (losses 1e-2, weights a few lr)."""
    c = load_berka()
    s = orc.make_schedule(c["T"], "cosine")
    p0 = orc.init_mlp_params(c["D"], c["d_layers"], seed=0)
    n = len(c["d_layers"])
    np.testing.assert_array_equal(flatten(p0, n).numpy()[:: c["stride"]], c["init_sample"])
    hist, p, ema = orc.train_loop(s, c["K"], c["Nn"], p0, c["batches"], float(c["lr"]), float(c["wd"]), c["steps"])
    ref = c["losses"]
    # The run is chaotic (see tests/test_gpu_train.py::test_berka_run_200_steps: a 1e-7 perturbation of the initial weights moves
    # the reference's own curve by up to 3.3e-2, mean 6.4e-3, and its weights by up to 1.6e-2): in a fresh process on the CPU that
    # made the fixture the oracle reproduces all of it bit for bit, but another thread count (set by an earlier test) or another
    # host's fp32 BLAS summation order is such a perturbation.  Bars: the first 10 losses tight, the rest within twice that floor.
    rel = np.abs(hist[:, 1] - ref[:, 1]) / np.abs(ref[:, 1])
    assert rel[:10].max() <= 1e-5, rel[:10]
    assert rel.max() <= 6.6e-2 and rel.mean() <= 1.3e-2, (rel.max(), rel.mean())
    assert np.all(hist[:, 0] == 0.0)
    err = np.abs(flatten(p, n).numpy()[:: c["stride"]] - c["final_sample"])
    assert err.max() <= 3.2e-2 and err.mean() <= 2e-3, (err.max(), err.mean())
    assert np.abs(flatten(ema, n).numpy()[:: c["stride"]] - c["ema_sample"]).max() <= 3.2e-2 * 0.2
