"""Generate tests/golden/*.npz by running the REFERENCE itself (container only).

Run:  python tools/make_golden.py
The reference is imported through tools/ref_shim.py from /root/reference (read-only).  Every
random draw the reference makes is captured by wrapping the torch RNG entry points it calls
(randint / randn / randn_like / rand_like / multinomial), so each fixture holds the inputs, the
injected noise and the reference's outputs.  Denoiser weights are NOT stored: they are rebuilt
from a seed by oracle.diffusion_oracle.init_mlp_params (plain torch CPU RNG) and loaded into the
reference module through load_state_dict, which keeps the fixtures small.
"""
from __future__ import annotations

import contextlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_shim  # noqa: E402

gmd, du, model = ref_shim.load()
from midst_toolkit.common.enumerations import IsTargetConditioned  # noqa: E402

from oracle import diffusion_oracle as orc  # noqa: E402  (only for init_mlp_params / names)

TF = ref_shim.load_tf()

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
CPU = torch.device("cpu")


@contextlib.contextmanager
def capture_rng(log: list):
    names = ["randint", "randn", "randn_like", "rand_like", "multinomial"]
    saved = {n: getattr(torch, n) for n in names}

    def wrap(n):
        def f(*a, **k):
            out = saved[n](*a, **k)
            log.append((n, out.clone()))
            return out
        return f

    for n in names:
        setattr(torch, n, wrap(n))
    try:
        yield
    finally:
        for n in names:
            setattr(torch, n, saved[n])


# extra constructor arguments of every diffusion built below (gen_parametrizations: the X0 / DIRECT variants, GMD:164-165)
DIFF_KW: dict = {}


def build(K, Nn, d_layers, T, sched="cosine", seed=0):
    K = np.asarray(K, dtype=np.int64)
    D = Nn + int(K.sum())
    params = orc.init_mlp_params(D, d_layers, seed=seed)
    dp = model.DiffusionParameters(layers_dimensions=list(d_layers), dropout=0.0)
    net = model.MLPDiffusion(D, 0, IsTargetConditioned.NONE, dp)
    net.load_state_dict(params, strict=True)
    net.eval()
    st = gmd.SchedulerType.COSINE if sched == "cosine" else gmd.SchedulerType.LINEAR
    diff = gmd.GaussianMultinomialDiffusion(K, Nn, net, num_timesteps=T, scheduler_type=st, device=CPU, **DIFF_KW)
    diff.eval()
    return diff, params


def synth_rows(B, K, Nn, g):
    x = torch.randn((B, Nn), generator=g).clamp(-5.2, 5.2)
    cols = [x]
    for k in K:
        if k > 0:
            cols.append(torch.randint(0, int(k), (B, 1), generator=g).float())
    return torch.cat(cols, dim=1)


def has_cat(K):
    return len(K) > 0 and int(K[0]) != 0


def gen_schedule(tag, T, sched):
    diff, _ = build([0], 2, [8], T, sched)
    d = {k: v.numpy() for k, v in diff.state_dict().items() if not k.startswith("_denoise_fn")}
    d["posterior_variance_f64"] = diff.posterior_variance.numpy()
    d["posterior_log_variance_clipped"] = diff.posterior_log_variance_clipped.numpy()
    d["posterior_mean_coef1"] = diff.posterior_mean_coef1.numpy()
    d["posterior_mean_coef2"] = diff.posterior_mean_coef2.numpy()
    mv = torch.cat([diff.posterior_variance[1].unsqueeze(0), (1.0 - diff.alphas)[1:]], dim=0)
    d["model_variance_f64"] = mv.numpy()
    d["model_log_variance_f64"] = torch.log(mv).numpy()
    np.savez_compressed(os.path.join(OUT, f"sched_{tag}.npz"), T=T, sched=sched, **d)


def gen_case(tag, K, Nn, d_layers, T, B, sched="cosine", P=4, T_chain=None):
    K = np.asarray(K, dtype=np.int64)
    C = len(K) if has_cat(K) else 0
    S = int(K.sum())
    g = torch.Generator().manual_seed(20250821)
    diff, params = build(K, Nn, d_layers, T, sched)
    x0 = synth_rows(B, K, Nn, g)
    out: dict[str, np.ndarray] = dict(K=K, Nn=Nn, d_layers=np.asarray(d_layers), T=T, sched=sched, x0=x0.numpy())

    # ---- mixed_loss, whole call with captured RNG, then replayed step by step for per-record rows
    log: list = []
    torch.manual_seed(1234)
    with capture_rng(log), torch.no_grad():
        m_mean, g_mean = diff.mixed_loss(x0, {"y": torch.zeros(B, dtype=torch.long)})
    t = log[0][1].clone()
    assert log[0][0] == "randint"
    # force the edge timesteps into the batch, then recompute everything with injected noise
    t[:4] = torch.tensor([0, 1, T - 1, 0])
    eps = log[1][1] if Nn > 0 else torch.zeros((B, 0))
    us = [e[1] for e in log[(2 if Nn > 0 else 1):]]
    u = torch.cat(us, dim=1) if C else torch.zeros((B, 0))
    assert u.shape[1] == (S if C else 0)
    with torch.no_grad():
        x_num_t = diff.gaussian_q_sample(x0[:, :Nn], t, noise=eps) if Nn > 0 else x0[:, :0]
        if C:
            log_x0 = du.index_to_log_onehot(x0[:, Nn:].long(), torch.from_numpy(K))
            it = iter(us)
            saved = torch.rand_like
            torch.rand_like = lambda ref: next(it)  # inject the captured uniforms, block by block
            try:
                log_xt = diff.q_sample(log_features_start=log_x0, timestep=t)
            finally:
                torch.rand_like = saved
            xt_idx = du.one_hot_encoding_to_categories(torch.exp(log_xt).round(), K)
            x_in = torch.cat([x_num_t, log_xt], dim=1)
        else:
            xt_idx = torch.zeros((B, 0), dtype=torch.int64)
            x_in = x_num_t
        m_out = diff._denoise_fn(x_in, t)
        lm_rows = torch.zeros((1,))
        lg_rows = torch.zeros((1,))
        if C:
            p_t = torch.ones_like(t).float() / T
            lm_rows = diff._multinomial_loss(m_out[:, Nn:], log_x0, log_xt, t, p_t) / len(K)
        if Nn > 0:
            lg_rows = diff._gaussian_loss(m_out[:, :Nn], x0[:, :Nn], x_num_t, t, eps)
    out.update(ml_t=t.numpy(), ml_eps=eps.numpy(), ml_u=u.numpy(), ml_x_num_t=x_num_t.numpy(),
               ml_xt_idx=xt_idx.numpy(), ml_model_in=x_in.numpy(), ml_model_out=m_out.numpy(),
               ml_lm_rows=lm_rows.numpy(), ml_lg_rows=lg_rows.numpy(),
               ml_ref_call_t=log[0][1].numpy(), ml_ref_call_means=np.array([m_mean.item(), g_mean.item()], dtype=np.float32))

    # ---- Tartan-Federer scoring (numeric block only, TF:40-95); the real reference function
    if Nn > 0:
        noise = torch.randn((P, Nn), generator=g)
        for k, (tt, add) in enumerate([(5, 0), (T // 2, 0), (T - 2, 1), (0, 0)]):
            if C:
                break  # TF feeds only the numeric block to the denoiser: valid only when S == 0
            with torch.no_grad():
                sc = TF.mixed_loss(diff, x0, {"y": torch.zeros(B * P, dtype=torch.long)}, noise.tolist(), P, add, tt)
            out[f"tf{k}_t"] = np.array([tt, add])
            out[f"tf{k}_score"] = sc.numpy()
        out["tf_noise"] = noise.numpy()

    # ---- single reverse steps with injected model_out / noise / uniforms (a16, a17)
    for k, tt in enumerate([T - 1, T // 2, 1, 0]):
        first = tt == T - 1
        z = torch.randn((B, Nn), generator=g) * 1.3
        z = z if first else z.double() + 1e-9 * torch.randn((B, Nn), generator=g, dtype=torch.float64)
        mo = torch.randn((B, Nn + S), generator=g) * 2.0
        n = torch.randn((B, Nn), generator=g, dtype=torch.float32 if first else torch.float64)
        tvec = torch.full((B,), tt, dtype=torch.long)
        saved_randn_like = torch.randn_like
        torch.randn_like = lambda ref: n
        try:
            with torch.no_grad():
                z_new = diff.gaussian_p_sample(mo[:, :Nn], z, tvec)["sample"] if Nn > 0 else z
        finally:
            torch.randn_like = saved_randn_like
        out[f"rs{k}_t"] = np.array(tt)
        out[f"rs{k}_z"] = z.numpy()
        out[f"rs{k}_model_out"] = mo.numpy()
        out[f"rs{k}_n"] = n.numpy()
        out[f"rs{k}_z_new"] = z_new.numpy()
        if C:
            z_idx = torch.stack([torch.randint(0, int(kk), (B,), generator=g) for kk in K], dim=1)
            u = torch.rand((B, S), generator=g)
            off = np.concatenate([[0], np.cumsum(K)])
            it = iter([u[:, off[c]:off[c + 1]] for c in range(C)])
            saved = torch.rand_like
            torch.rand_like = lambda ref: next(it)
            try:
                with torch.no_grad():
                    log_z = du.index_to_log_onehot(z_idx, torch.from_numpy(K))
                    logp = diff.p_pred(mo[:, Nn:], log_z, tvec)
                    log_z_new = diff.log_sample_categorical(logp)
            finally:
                torch.rand_like = saved
            out[f"rs{k}_z_idx"] = z_idx.numpy()
            out[f"rs{k}_u"] = u.numpy()
            out[f"rs{k}_logp"] = logp.numpy()
            out[f"rs{k}_z_idx_new"] = du.one_hot_encoding_to_categories(torch.exp(log_z_new).round(), K).numpy()

    # ---- a full reverse chain on a short schedule, every draw captured (a18)
    if T_chain:
        diff2, _ = build(K, Nn, d_layers, T_chain, sched)
        Bc = min(B, 64)
        log2: list = []
        torch.manual_seed(4321)
        with capture_rng(log2):
            smp, outs = diff2.sample(Bc, torch.ones(1))
        kinds = [e[0] for e in log2]
        assert kinds[0] == "randn"
        pos = 1
        out["ch_z0"] = log2[0][1].numpy()
        if C:
            out["ch_u0"] = torch.cat([e[1] for e in log2[pos:pos + C]], dim=1).numpy()
            pos += C
        assert kinds[pos] == "multinomial"
        pos += 1
        ns, uss = [], []
        for i in range(T_chain):
            assert kinds[pos] == "randn_like", (i, kinds[pos])
            ns.append(log2[pos][1].double().numpy())  # fp32 on the first step, fp64 after: stored as fp64
            pos += 1
            if C:
                uss.append(torch.cat([e[1] for e in log2[pos:pos + C]], dim=1).numpy())
                pos += C
        assert pos == len(log2)
        out["ch_T"] = np.array(T_chain)
        out["ch_noise"] = np.stack(ns)
        out["ch_first_noise_is_f32"] = np.array(log2[1 + (C if C else 0) + 1][1].dtype == torch.float32)
        if C:
            out["ch_u"] = np.stack(uss)
        out["ch_sample"] = smp.numpy()
        assert smp.dtype == torch.float64

    np.savez_compressed(os.path.join(OUT, f"case_{tag}.npz"), **out)
    print(f"case {tag}: B={B} D={Nn + S} means={m_mean.item():.6f},{g_mean.item():.6f}")


def gen_train(tag, K, Nn, d_layers, T, B, steps, lr, wd, n_rows=99):
    """The reference's ClavaDDPMTrainer (trainer.py) for `steps` iterations on a fixed synthetic table, every
    random draw captured per iteration.  Stored: the batches, the draws, the two loss means per iteration and
    the trained / EMA weights (flattened in oracle.mlp_param_names order, weight then bias)."""
    import importlib

    trainer_mod = importlib.import_module("midst_toolkit.models.clavaddpm.trainer")
    K = np.asarray(K, dtype=np.int64)
    C = len(K) if has_cat(K) else 0
    S = int(K.sum())
    g = torch.Generator().manual_seed(777)
    diff, params = build(K, Nn, d_layers, T)
    diff.train()
    table = synth_rows(n_rows, K, Nn, g)
    order = [torch.randint(0, n_rows, (B,), generator=g) for _ in range(steps)]

    def batches():
        for idx in order:
            yield table[idx], torch.zeros(B, dtype=torch.long)

    tr = trainer_mod.ClavaDDPMTrainer(diff, batches(), lr, wd, steps, device=CPU)
    losses, draws = [], []
    inner = tr._train_step

    def spy(x, y):
        log: list = []
        with capture_rng(log):
            lm, lg = inner(x, y)
        losses.append([float(lm), float(lg)])
        draws.append(log)
        return lm, lg

    tr._train_step = spy
    torch.manual_seed(99)
    tr.train()
    ts, epss, us = [], [], []
    for log in draws:
        assert log[0][0] == "randint"
        ts.append(log[0][1].numpy())
        pos = 1
        if Nn > 0:
            assert log[pos][0] == "randn_like"
            epss.append(log[pos][1].numpy())
            pos += 1
        if C:
            us.append(torch.cat([e[1] for e in log[pos:pos + C]], dim=1).numpy())
            pos += C
        assert pos == len(log), (pos, len(log))
    names = orc.mlp_param_names(len(d_layers))
    sd, ema_sd = diff._denoise_fn.state_dict(), tr.ema_model.state_dict()
    flat = lambda d: np.concatenate([np.concatenate([d[n + ".weight"].detach().numpy().ravel(), d[n + ".bias"].detach().numpy().ravel()]) for n in names])
    np.savez_compressed(
        os.path.join(OUT, f"train_{tag}.npz"), K=K, Nn=Nn, d_layers=np.asarray(d_layers), T=T, sched="cosine", lr=lr, wd=wd, steps=steps,
        x=np.stack([table[i].numpy() for i in order]), t=np.stack(ts), eps=np.stack(epss) if epss else np.zeros((steps, B, 0), np.float32),
        u=np.stack(us) if us else np.zeros((steps, B, 0), np.float32), losses=np.asarray(losses, dtype=np.float64),
        final=flat(sd).astype(np.float32), ema=flat(ema_sd).astype(np.float32))
    print(f"train {tag}: losses {losses[0]} -> {losses[-1]}")


def gen_train_berka(steps=200, B=24, lr=6e-4, wd=1e-5):
    """BASELINE configs[0] / SURVEY 8c: the reference trainer for 200 iterations on the Berka 'trans' subset the reference's own
    integration test trains on (tests/integration/assets/single_table/train.csv, 99 rows x 8 columns, all in the numeric block
    as the shipped pipeline puts them), normalised by the reference's `normalize` (QuantileTransformer, output normal), with
    the test's denoiser [512, 1024, 1024, 1024, 1024, 512], T = 1000, batch 24.  Stored: the already-transformed matrix, the
    batch order, every draw, the two loss means per iteration, and - the trained image is 4.7 M parameters - a strided
    sample (every 997th entry) of the trained and EMA weights."""
    import importlib

    import pandas as pd
    from midst_toolkit.common.dataset_transformations import normalize
    from midst_toolkit.common.enumerations import DataSplit, Normalization

    trainer_mod = importlib.import_module("midst_toolkit.models.clavaddpm.trainer")
    raw = pd.read_csv("/root/reference/tests/integration/assets/single_table/train.csv").to_numpy(dtype=np.float64)
    data, _ = normalize({DataSplit.TRAIN.value: raw}, Normalization.QUANTILE, seed=0)
    table = torch.from_numpy(data[DataSplit.TRAIN.value].astype(np.float32))
    n_rows, Nn = table.shape
    d_layers = [512, 1024, 1024, 1024, 1024, 512]
    g = torch.Generator().manual_seed(778)
    diff, params = build([0], Nn, d_layers, 1000)
    diff.train()
    order = [torch.randint(0, n_rows, (B,), generator=g) for _ in range(steps)]

    def batches():
        for idx in order:
            yield table[idx], torch.zeros(B, dtype=torch.long)

    tr = trainer_mod.ClavaDDPMTrainer(diff, batches(), lr, wd, steps, device=CPU)
    losses, draws = [], []
    inner = tr._train_step

    def spy(x, y):
        log: list = []
        with capture_rng(log):
            lm, lg = inner(x, y)
        losses.append([float(lm), float(lg)])
        draws.append(log)
        return lm, lg

    tr._train_step = spy
    torch.manual_seed(100)
    tr.train()
    ts = np.stack([log[0][1].numpy() for log in draws])
    epss = np.stack([log[1][1].numpy() for log in draws])
    assert all(len(log) == 2 and log[0][0] == "randint" and log[1][0] == "randn_like" for log in draws)
    names = orc.mlp_param_names(len(d_layers))
    sd, ema_sd = diff._denoise_fn.state_dict(), tr.ema_model.state_dict()
    flat = lambda d: np.concatenate([np.concatenate([d[n + ".weight"].detach().numpy().ravel(), d[n + ".bias"].detach().numpy().ravel()]) for n in names])
    np.savez_compressed(
        os.path.join(OUT, "train_berka.npz"), K=np.array([0]), Nn=Nn, d_layers=np.asarray(d_layers), T=1000, sched="cosine", lr=lr, wd=wd, steps=steps,
        table=table.numpy(), order=np.stack([o.numpy() for o in order]), t=ts, eps=epss, losses=np.asarray(losses, dtype=np.float64),
        stride=997, final_sample=flat(sd)[::997].astype(np.float32), ema_sample=flat(ema_sd)[::997].astype(np.float32),
        init_sample=flat(params)[::997].astype(np.float32))
    print(f"train berka: {n_rows} x {Nn}, losses {losses[0]} -> {np.mean([l[1] for l in losses[-20:]]):.4f} (mean of the last 20)")


def guidance_function(centers: torch.Tensor, scale: float, T: int):
    """A closed-form stand-in for the classifier gradient of SYNTH:429-451 (same call signature and dtype): the
    gradient of log N(features; centers[y], I) scaled by a factor that grows with t."""
    def conditioning_function(features, timestep, **kwargs):
        y = kwargs["y"].view(-1).long()
        x = features.detach().float()
        return (centers[y] - x) * (scale * (1.0 + timestep.float()[:, None] / T))
    return conditioning_function


def gen_guided(tag, Nn, d_layers, T_chain, B, n_classes=3, scale=0.35):
    """conditional_sample (GMD:1319) and sample(..., conditioning_function) (GMD:1370) of the reference with every
    draw captured; numeric-only table, as in the multi-table pipeline that uses them."""
    diff, params = build([0], Nn, d_layers, T_chain)
    g = torch.Generator().manual_seed(99)
    centers = torch.randn((n_classes, Nn), generator=g)
    targets = torch.randint(0, n_classes, (B, 1), generator=g)
    cond = guidance_function(centers, scale, T_chain)
    out = {"Nn": np.array(Nn), "d_layers": np.array(d_layers), "T": np.array(T_chain), "centers": centers.numpy(),
           "scale": np.array(scale), "targets": targets.numpy()}
    out.update({f"w_{k}": v.numpy() for k, v in params.items()})
    for name in ("cs", "sg"):
        log: list = []
        torch.manual_seed(777 if name == "cs" else 778)
        with capture_rng(log):
            if name == "cs":
                with torch.no_grad():
                    smp, _ = diff.conditional_sample(targets, model_kwargs={"y": targets}, conditioning_function=cond)
            else:
                smp, outs = diff.sample(B, torch.ones(n_classes), model_kwargs={"y": targets}, conditioning_function=cond)
        kinds = [e[0] for e in log]
        assert kinds[0] == "randn"
        pos = 1
        if name == "sg":
            assert kinds[pos] == "multinomial"
            out["sg_y"] = outs["y"].numpy()
            pos += 1
        assert kinds[pos:] == ["randn_like"] * T_chain, kinds
        out[f"{name}_z0"] = log[0][1].numpy()
        out[f"{name}_noise"] = np.stack([e[1].double().numpy() for e in log[pos:]])
        out[f"{name}_sample"] = smp.numpy()
        assert smp.dtype == torch.float64
    np.savez_compressed(os.path.join(OUT, f"guided_{tag}.npz"), **out)
    print(f"guided {tag}: B={B} Nn={Nn} T={T_chain}")


def gen_classifier(tag="A_numeric", Nn=8, d_layers=(64, 64), T_chain=12, B=48, n_classes=5, hidden=(64, 128, 96), scale=1.5):
    """The reference's own guidance classifier (MODEL:45-120) in eval mode, as the synthesizer uses it (TRAIN:447), through the
    reference's own conditioning function (SYNTH:429-451): logits and guidance gradients on fixed inputs (per-row timesteps,
    with and without remove_first_col), and conditional_sample (GMD:1319) driven by it with every draw captured.  BatchNorm
    running statistics and affine parameters are randomised so that folding them is not a no-op.  The nn.Transformer the
    reference constructs and never calls is left out of the stored state dict (4 M parameters)."""
    from midst_toolkit.models.clavaddpm import synthesizer as synth

    torch.manual_seed(4242)
    clf = model.Classifier(input_dimension=Nn, output_dimension=n_classes, timestep_dimension=128, hidden_sizes=list(hidden), dropout_prob=0.1)
    g = torch.Generator().manual_seed(17)
    for m in clf.model:
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.3)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 1.5 + 0.25)
            with torch.no_grad():
                m.weight.copy_(1.0 + 0.3 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
    clf.eval()
    out = {"Nn": np.array(Nn), "d_layers": np.array(d_layers), "T": np.array(T_chain), "n_classes": np.array(n_classes),
           "hidden": np.array(hidden), "scale": np.array(scale)}
    out.update({f"c_{k}": v.numpy() for k, v in clf.state_dict().items() if not k.startswith("transformer_layer.")})
    # ---- forward and guidance gradient on fixed inputs, per-row timesteps over the whole table
    Bx = 200
    x = torch.randn((Bx, Nn), generator=g) * 1.2
    t = torch.randint(0, 1000, (Bx,), generator=g)
    t[:3] = torch.tensor([0, 1, 999])
    y = torch.randint(0, n_classes, (Bx, 1), generator=g)
    with torch.no_grad():
        logits = clf(x, t)
    cond = synth._get_conditioning_function(clf, scale)
    grad = cond(x.double(), t, y=y)                      # the chain hands fp64 features after the first step
    x1 = torch.cat([torch.randn((Bx, 1), generator=g), x], 1)
    grad_rfc = cond(x1, t, y=y, remove_first_col=True)
    out.update(x=x.numpy(), t=t.numpy(), y=y.numpy(), logits=logits.numpy(), grad=grad.numpy(), x_rfc=x1.numpy(), grad_rfc=grad_rfc.numpy())
    # ---- conditional_sample guided by the classifier
    diff, params = build([0], Nn, list(d_layers), T_chain)
    out.update({f"w_{k}": v.numpy() for k, v in params.items()})
    targets = torch.randint(0, n_classes, (B, 1), generator=g)
    log: list = []
    torch.manual_seed(779)
    with capture_rng(log):
        with torch.no_grad():
            smp, _ = diff.conditional_sample(targets, model_kwargs={"y": targets}, conditioning_function=cond)
    kinds = [e[0] for e in log]
    assert kinds == ["randn"] + ["randn_like"] * T_chain, kinds
    out.update(targets=targets.numpy(), cs_z0=log[0][1].numpy(), cs_noise=np.stack([e[1].double().numpy() for e in log[1:]]), cs_sample=smp.numpy())
    np.savez_compressed(os.path.join(OUT, f"classifier_{tag}.npz"), **out)
    print(f"classifier {tag}: logits {tuple(logits.shape)} |grad| {float(grad.abs().mean()):.3e} sample {tuple(smp.shape)}")


def gen_parametrizations():
    """GaussianParametrization.X0 and Parametrization.DIRECT (GMD:399-408, 822-832; no reference config uses them): the same
    fixtures as for the defaults, from the reference constructed with both - a case file (mixed_loss rows, single reverse steps
    with the reference's p_pred, a whole chain), the KL Gaussian loss on that case, and a DDIM chain."""
    DIFF_KW.update(gaussian_parametrization=gmd.GaussianParametrization.X0, parametrization=gmd.Parametrization.DIRECT)
    try:
        gen_case("P_x0direct", [3, 7, 10, 2, 12], 6, [128, 128], 1000, 96, T_chain=16)
        gen_kl("P_x0direct")
        gen_ddim("P_x0direct", [3, 7, 10, 2, 12], 6, [128, 128], 16, 64)
        gen_ddim_reverse("P_x0direct", 6, [64, 64], 12, 48)
    finally:
        DIFF_KW.clear()


def gen_ddim_reverse(tag, Nn, d_layers, T, B):
    """gaussian_ddim_reverse_step (GMD:1160-1186) at a few timesteps with injected denoiser output."""
    diff, _ = build([0], Nn, d_layers, T)
    g = torch.Generator().manual_seed(31)
    out = {"Nn": np.array(Nn), "T": np.array(T), "d_layers": np.array(d_layers)}
    for k, tt in enumerate([0, T // 2, T - 1]):
        z = torch.randn((B, Nn), generator=g)
        mo = torch.randn((B, Nn), generator=g)
        with torch.no_grad():
            zn = diff.gaussian_ddim_reverse_step(mo, z, torch.full((B,), tt, dtype=torch.long))
        out[f"r{k}_t"] = np.array(tt)
        out[f"r{k}_z"], out[f"r{k}_model_out"], out[f"r{k}_z_new"] = z.numpy(), mo.numpy(), zn.numpy()
    np.savez_compressed(os.path.join(OUT, f"ddimrev_{tag}.npz"), **out)
    print(f"ddimrev {tag}: B={B}")


def gen_ddim(tag, K, Nn, d_layers, T_chain, B, sched="cosine"):
    """sample_ddim (GMD:1259) of the reference with every draw captured."""
    K = np.asarray(K, dtype=np.int64)
    C = len(K) if has_cat(K) else 0
    diff, _ = build(K, Nn, d_layers, T_chain, sched)
    log: list = []
    torch.manual_seed(5150)
    with capture_rng(log):
        smp, outs = diff.sample_ddim(B, torch.ones(1))
    kinds = [e[0] for e in log]
    assert kinds[0] == "randn"
    out = {"K": K, "Nn": np.array(Nn), "d_layers": np.array(d_layers), "T": np.array(T_chain), "sched": np.array(sched),
           "z0": log[0][1].numpy()}
    pos = 1
    if C:
        out["u0"] = torch.cat([e[1] for e in log[pos:pos + C]], dim=1).numpy()
        pos += C
    assert kinds[pos] == "multinomial"
    pos += 1
    ns, uss = [], []
    for i in range(T_chain):
        assert kinds[pos] == "randn_like", (i, kinds[pos])
        ns.append(log[pos][1].numpy())
        pos += 1
        if C:
            uss.append(torch.cat([e[1] for e in log[pos:pos + C]], dim=1).numpy())
            pos += C
    assert pos == len(log)
    out["noise"] = np.stack(ns)
    if C:
        out["u"] = np.stack(uss)
    out["sample"] = smp.numpy()
    out["sample_dtype"] = np.array(str(smp.dtype))
    np.savez_compressed(os.path.join(OUT, f"ddim_{tag}.npz"), **out)
    print(f"ddim {tag}: B={B} dtype={smp.dtype}")


def gen_ddim_family(tag="A_numeric", Nn=8, d_layers=(64, 64), T_chain=12, B=48, n_classes=3, scale=0.35):
    """The rest of the reference's DDIM family, every draw captured: gaussian_ddim_sample with eta > 0 (GMD:1119), the same with
    score-based guidance (condition_score, GMD:617-661), sample_ddim with a conditioning function (GMD:1259, eta = 0),
    gaussian_ddim_reverse_sample (GMD:1189), and multinomial_ddim_step with eta > 0 (GMD:1218) on a ragged categorical table."""
    diff, params = build([0], Nn, list(d_layers), T_chain)
    g = torch.Generator().manual_seed(4100)
    centers = torch.randn((n_classes, Nn), generator=g)
    targets = torch.randint(0, n_classes, (B, 1), generator=g)
    cond = guidance_function(centers, scale, T_chain)
    z0 = torch.randn((B, Nn), generator=g)
    x_data = torch.randn((B, Nn), generator=g) * 0.8
    out = {"Nn": np.array(Nn), "d_layers": np.array(d_layers), "T": np.array(T_chain), "centers": centers.numpy(), "scale": np.array(scale),
           "targets": targets.numpy(), "z0": z0.numpy(), "x_data": x_data.numpy(), "eta": np.array(0.6), "eta_cs": np.array(0.4)}
    out.update({f"w_{k}": v.numpy() for k, v in params.items()})
    for name, kw in (("eta", dict(eta=0.6)), ("etacs", dict(eta=0.4, model_kwargs={"y": targets}, conditioning_function=cond))):
        log: list = []
        torch.manual_seed(31 if name == "eta" else 32)
        with capture_rng(log):
            smp = diff.gaussian_ddim_sample(z0.clone(), T_chain, {}, **kw)
        assert [e[0] for e in log] == ["randn_like"] * T_chain
        out[f"{name}_noise"] = np.stack([e[1].numpy() for e in log])
        out[f"{name}_sample"] = smp.numpy()
        assert smp.dtype == torch.float32
    log = []
    torch.manual_seed(33)
    with capture_rng(log):
        smp, outs = diff.sample_ddim(B, torch.ones(n_classes), model_kwargs={"y": targets}, conditioning_function=cond)
    kinds = [e[0] for e in log]
    assert kinds == ["randn", "multinomial"] + ["randn_like"] * T_chain, kinds
    out.update(sd_z0=log[0][1].numpy(), sd_y=outs["y"].numpy(), sd_sample=smp.numpy())
    out["rev_sample"] = diff.gaussian_ddim_reverse_sample(x_data.clone(), T_chain, {}).numpy()
    # multinomial_ddim_step with eta > 0, ragged categorical blocks
    K = np.asarray([3, 7, 10, 2, 12], dtype=np.int64)
    diffc, _ = build(K, 6, [128, 128], 16)
    S = int(K.sum())
    Bc = 64
    oc = torch.randn((Bc, S), generator=g) * 1.5
    zi = torch.stack([torch.randint(0, int(k), (Bc,), generator=g) for k in K], dim=1)
    log_z = orc.log_onehot(zi, K)
    md = {"md_K": K, "md_T": np.array(16), "md_out_cat": oc.numpy(), "md_z_idx": zi.numpy(), "md_eta": np.array(0.5)}
    for t in (0, 1, 7, 15):
        log = []
        torch.manual_seed(40 + t)
        with capture_rng(log):
            res = diffc.multinomial_ddim_step(oc, log_z, torch.full((Bc,), t, dtype=torch.long), eta=0.5)
        md[f"md_u_{t}"] = torch.cat([e[1] for e in log], dim=1).numpy()
        md[f"md_idx_{t}"] = one_hot_to_idx(res, K)
    out.update(md)
    np.savez_compressed(os.path.join(OUT, f"ddimx_{tag}.npz"), **out)
    print(f"ddim family {tag}: eta sample {tuple(out['eta_sample'].shape)}, reverse |x| {float(np.abs(out['rev_sample']).mean()):.3f}")


def one_hot_to_idx(log_x: torch.Tensor, K) -> np.ndarray:
    off = np.concatenate([[0], np.cumsum(K)])
    return np.stack([log_x[:, off[k]:off[k + 1]].argmax(dim=1).numpy() for k in range(len(K))], axis=1)


def gen_edm(tag="latent24", d_in=24, dim_t=128, B=96, num_steps=10):
    """TabSyn's EDM denoiser (models/tabsyn/model/modules.py, utils.py) run by the reference: MLPDiffusion.forward, Preconditioner.forward
    (per-row and scalar sigma), EDMLoss on captured draws, Model.forward, and the Euler-Heun sampler with every draw captured."""
    from midst_toolkit.models.tabsyn.model import modules as tm
    from midst_toolkit.models.tabsyn.model import utils as tu

    torch.manual_seed(606)
    net = tm.MLPDiffusion(d_in, dim_t)
    mdl = tm.Model(net)
    g = torch.Generator().manual_seed(607)
    x = torch.randn((B, d_in), generator=g)
    labels = torch.randn((B,), generator=g) * 0.8
    sigma = (torch.randn((B,), generator=g) * 1.2 - 1.2).exp()
    out = {"d_in": np.array(d_in), "dim_t": np.array(dim_t), "num_steps": np.array(num_steps), "x": x.numpy(), "labels": labels.numpy(),
           "sigma": sigma.numpy()}
    out.update({f"w_{k}": v.numpy() for k, v in net.state_dict().items()})
    with torch.no_grad():
        out["forward"] = net(x, labels).numpy()
        out["denoise"] = mdl.preconditioner(x, sigma).numpy()
        out["denoise_scalar"] = mdl.preconditioner(x, torch.tensor(0.37)).numpy()
        log: list = []
        torch.manual_seed(608)
        with capture_rng(log):
            loss = mdl.loss_fn(mdl.preconditioner, x)
        assert [e[0] for e in log] == ["randn", "randn_like"]
        out.update(loss_rnd=log[0][1].numpy(), loss_noise=log[1][1].numpy(), loss=loss.numpy(), loss_mean=np.array(float(loss.mean(-1).mean())))
        log = []
        torch.manual_seed(609)
        import torch as _t
        saved = tu.randn_like
        tu.randn_like = lambda v: (log.append(("randn_like", saved(v))) or log[-1][1])
        try:
            with capture_rng(log):
                smp = tu.sample(mdl.preconditioner, B, d_in, num_steps=num_steps, device=CPU)
        finally:
            tu.randn_like = saved
        kinds = [e[0] for e in log]
        assert kinds == ["randn"] + ["randn_like"] * num_steps, kinds
        out.update(s_latents=log[0][1].numpy(), s_noise=np.stack([e[1].numpy() for e in log[1:]]), s_sample=smp.numpy())
    np.savez_compressed(os.path.join(OUT, f"edm_{tag}.npz"), **out)
    print(f"edm {tag}: loss mean {float(out['loss_mean']):.4f}, sample |x| {float(np.abs(out['s_sample']).mean()):.3f}")


def gen_label_embedding(tag="label3", K=(3, 4), Nn=3, d_layers=(64, 64), T=50, B=48, n_labels=3, T_chain=8):
    """IsTargetConditioned.EMBEDDING with classes (MODEL:583-585, 610-612): the reference's denoiser forward with and without
    ``y``, its mixed_loss rows with injected noise and ``y`` (GMD:1042), and a whole ``sample`` chain whose labels are drawn from
    ``target_dist`` (GMD:1401) - every draw captured."""
    K = np.asarray(K, dtype=np.int64)
    C, S = len(K), int(K.sum())
    D = Nn + S
    g = torch.Generator().manual_seed(77)
    params = orc.init_mlp_params(D, list(d_layers), seed=9)
    params["label_embedding.weight"] = torch.randn((n_labels, 128), generator=g)

    def make(T_):
        dp = model.DiffusionParameters(layers_dimensions=list(d_layers), dropout=0.0)
        net = model.MLPDiffusion(D, n_labels, IsTargetConditioned.EMBEDDING, dp)
        net.load_state_dict(params, strict=True)
        net.eval()
        diff = gmd.GaussianMultinomialDiffusion(K, Nn, net, num_timesteps=T_, scheduler_type=gmd.SchedulerType.COSINE, device=CPU)
        diff.eval()
        return diff

    diff = make(T)
    out: dict[str, np.ndarray] = dict(K=K, Nn=Nn, d_layers=np.asarray(d_layers), T=T, n_labels=n_labels)
    out.update({"p_" + k: v.numpy() for k, v in params.items()})
    # ---- denoiser forward
    x = torch.randn((B, D), generator=g)
    t = torch.randint(0, T, (B,), generator=g)
    y = torch.randint(0, n_labels, (B, 1), generator=g)  # [B, 1]: MODEL:611 squeezes
    with torch.no_grad():
        out.update(fw_x=x.numpy(), fw_t=t.numpy(), fw_y=y.numpy(), fw_out_y=diff._denoise_fn(x, t, y).numpy(),
                   fw_out_none=diff._denoise_fn(x, t).numpy())
    # ---- mixed_loss: one whole call with captured draws, then the per-record rows from the injected pieces
    x0 = synth_rows(B, K, Nn, g)
    yl = torch.randint(0, n_labels, (B,), generator=g)
    log: list = []
    torch.manual_seed(99)
    with capture_rng(log), torch.no_grad():
        m_mean, g_mean = diff.mixed_loss(x0, {"y": yl})
    t = log[0][1].clone()
    eps = log[1][1]
    us = [e[1] for e in log[2:]]
    with torch.no_grad():
        x_num_t = diff.gaussian_q_sample(x0[:, :Nn], t, noise=eps)
        log_x0 = du.index_to_log_onehot(x0[:, Nn:].long(), torch.from_numpy(K))
        it = iter(us)
        saved = torch.rand_like
        torch.rand_like = lambda ref: next(it)
        try:
            log_xt = diff.q_sample(log_features_start=log_x0, timestep=t)
        finally:
            torch.rand_like = saved
        m_out = diff._denoise_fn(torch.cat([x_num_t, log_xt], dim=1), t, yl)
        lm_rows = diff._multinomial_loss(m_out[:, Nn:], log_x0, log_xt, t, torch.ones_like(t).float() / T) / C
        lg_rows = diff._gaussian_loss(m_out[:, :Nn], x0[:, :Nn], x_num_t, t, eps)
    assert abs(float(lm_rows.mean()) - float(m_mean)) < 1e-6 and abs(float(lg_rows.mean()) - float(g_mean)) < 1e-6
    out.update(ml_x0=x0.numpy(), ml_y=yl.numpy(), ml_t=t.numpy(), ml_eps=eps.numpy(), ml_u=torch.cat(us, dim=1).numpy(),
               ml_lm_rows=lm_rows.numpy(), ml_lg_rows=lg_rows.numpy(), ml_means=np.array([m_mean.item(), g_mean.item()], dtype=np.float32))
    # ---- sample: labels drawn from target_dist
    diff2 = make(T_chain)
    Bc = 32
    log2: list = []
    torch.manual_seed(4321)
    with capture_rng(log2):
        smp, outs = diff2.sample(Bc, torch.tensor([0.2, 0.5, 0.3]))
    kinds = [e[0] for e in log2]
    assert kinds[0] == "randn" and kinds[1 + C] == "multinomial"
    out["ch_z0"] = log2[0][1].numpy()
    out["ch_u0"] = torch.cat([e[1] for e in log2[1:1 + C]], dim=1).numpy()
    out["ch_y"] = log2[1 + C][1].numpy()
    assert torch.equal(outs["y"].cpu().long(), log2[1 + C][1].long())
    pos = 2 + C
    ns, uss = [], []
    for i in range(T_chain):
        assert kinds[pos] == "randn_like"
        ns.append(log2[pos][1].double().numpy())
        uss.append(torch.cat([e[1] for e in log2[pos + 1:pos + 1 + C]], dim=1).numpy())
        pos += 1 + C
    assert pos == len(log2)
    out.update(ch_T=np.array(T_chain), ch_noise=np.stack(ns), ch_u=np.stack(uss), ch_sample=smp.numpy())
    np.savez_compressed(os.path.join(OUT, f"label_{tag}.npz"), **out)
    print(f"label {tag}: B={B} D={D} means={m_mean.item():.6f},{g_mean.item():.6f}")


def gen_loader():
    """Batches of the reference's FastTensorDataLoader over two shuffled epochs (data_loaders.py:486-551)."""
    from midst_toolkit.models.clavaddpm.data_loaders import FastTensorDataLoader

    g = torch.Generator().manual_seed(5)
    x = torch.randn((37, 3), generator=g)
    y = torch.arange(37).float()
    torch.manual_seed(2024)
    dl = FastTensorDataLoader([x, y], batch_size=8, shuffle=True)
    out = {"x": x.numpy(), "y": y.numpy(), "n_batches": np.array(len(dl))}
    k = 0
    for _ in range(2):
        for bx, by in dl:
            out[f"b{k}_x"], out[f"b{k}_y"] = bx.numpy(), by.numpy()
            k += 1
    out["count"] = np.array(k)
    np.savez_compressed(os.path.join(OUT, "loader.npz"), **out)
    print("loader:", k, "batches")


def gen_quantile():
    """The reference's own quantile normaliser (common/dataset_transformations.py:46-84, Normalization.QUANTILE) fitted
    on its single-table asset (99 rows -> 10 quantiles) and on a 36,000-row synthetic split (-> 1000 quantiles); the
    fixture holds the fitted tables, inputs in the normalised space (chain-like values, both bound regions, saturating values, NaN)
    and the fitted object's inverse_transform of them (what synthesizer.py:277 computes)."""
    import pandas as pd
    from midst_toolkit.common.dataset_transformations import normalize
    from midst_toolkit.common.enumerations import Normalization

    g = np.random.default_rng(77)
    csv = pd.read_csv("/root/reference/tests/integration/assets/single_table/train.csv")
    small = csv.select_dtypes("number").to_numpy(dtype=np.float64)
    big = np.stack([g.lognormal(3.0, 1.2, 36000), np.round(g.normal(0, 40, 36000)), g.integers(0, 7, 36000).astype(np.float64),
                    g.exponential(2.0, 36000) * (g.random(36000) > 0.3), g.normal(1e6, 5e3, 36000)], axis=1)
    for tag, train in (("small", small), ("big", big)):
        _, qt = normalize({"train": train}, Normalization.QUANTILE, seed=0)
        n = 4096
        x = g.normal(0, 1.3, (n, train.shape[1]))
        x[:64] = g.normal(0, 1, (64, train.shape[1])) * 0.02 + np.where(g.random((64, train.shape[1])) > 0.5, 5.2, -5.2)
        x[64, :] = 40.0  # Phi saturates to exactly 1 / 0 (sklearn refuses infinities)
        x[65, :] = -40.0
        x[66, 0] = np.nan
        x[67, :] = 0.0
        x[68:100] = np.sign(x[68:100]) * (5.0 + 4.0 * g.random((32, train.shape[1])))
        want = qt.inverse_transform(x.copy())
        np.savez_compressed(os.path.join(OUT, f"quantile_{tag}.npz"), x=x, quantiles=qt.quantiles_, references=qt.references_,
                            want=want, output_normal=np.array(int(qt.output_distribution == "normal")))
        print(f"quantile {tag}: n_quantiles={qt.n_quantiles_} columns={train.shape[1]}")


def gen_kl(tag, out_tag=None):
    """The KL Gaussian loss (gaussian_loss_type=KL: GMD:431-484 _vb_terms_bpd, DU:9-91) on the inputs the case fixture
    already holds (x0, t, x_t, model_out of the reference's own mixed_loss replay): per-record values for the stored t,
    for t = 0 everywhere (decoder NLL) and t = 1 everywhere, and the two means of a whole mixed_loss call."""
    z = np.load(os.path.join(OUT, f"case_{tag}.npz"))
    K, Nn, T, d_layers, sched = z["K"], int(z["Nn"]), int(z["T"]), list(z["d_layers"]), str(z["sched"])
    diff, _ = build(K, Nn, d_layers, T, sched)
    diff.gaussian_loss_type = gmd.GaussianLossType.KL
    x0 = torch.from_numpy(z["x0"])
    B = x0.shape[0]
    mo = torch.from_numpy(z["ml_model_out"])[:, :Nn]
    xt = torch.from_numpy(z["ml_x_num_t"])
    t = torch.from_numpy(z["ml_t"])
    eps = torch.from_numpy(z["ml_eps"])
    out = {}
    with torch.no_grad():
        for name, tv in (("t", t), ("t0", torch.zeros_like(t)), ("t1", torch.ones_like(t))):
            rows = diff._gaussian_loss(mo, x0[:, :Nn], xt, tv, eps)
            assert rows.dtype == torch.float64 and rows.shape == (B,)
            out[f"kl_rows_{name}"] = rows.numpy()
        log: list = []
        torch.manual_seed(1234)
        with capture_rng(log):
            m_mean, g_mean = diff.mixed_loss(x0, {"y": torch.zeros(B, dtype=torch.long)})
    out["call_t"] = log[0][1].numpy()
    out["call_eps"] = log[1][1].numpy()
    out["call_means"] = np.array([float(m_mean), float(g_mean)], dtype=np.float64)
    out["call_gauss_dtype"] = np.array(str(g_mean.dtype))
    np.savez_compressed(os.path.join(OUT, f"kl_{out_tag or tag}.npz"), **out)
    print(f"kl {out_tag or tag}: rows dtype float64, mixed_loss gauss mean {float(g_mean):.6g} ({g_mean.dtype})")


def gen_affine():
    """The reference's STANDARD and MINMAX normalisers (common/dataset_transformations.py:66-69) fitted on its single-table
    asset; inputs in the normalised space and the fitted objects' inverse_transform of them."""
    import pandas as pd
    from midst_toolkit.common.dataset_transformations import normalize
    from midst_toolkit.common.enumerations import Normalization

    g = np.random.default_rng(78)
    train = pd.read_csv("/root/reference/tests/integration/assets/single_table/train.csv").select_dtypes("number").to_numpy(dtype=np.float64)
    out = {}
    for tag, norm in (("standard", Normalization.STANDARD), ("minmax", Normalization.MINMAX)):
        _, sc = normalize({"train": train}, norm, seed=0)
        x = g.normal(0, 1.5, (512, train.shape[1])) if tag == "standard" else g.random((512, train.shape[1])) * 1.4 - 0.2
        out[f"{tag}_x"] = x
        out[f"{tag}_want"] = sc.inverse_transform(x.copy())
        if tag == "standard":
            out["standard_scale"], out["standard_mean"] = sc.scale_, sc.mean_
        else:
            out["minmax_min"], out["minmax_scale"] = sc.min_, sc.scale_
    np.savez_compressed(os.path.join(OUT, "affine.npz"), **out)
    print("affine: standard + minmax,", train.shape[1], "columns")


def main():
    if "--affine-only" in sys.argv:
        gen_affine()
        return
    if "--param-only" in sys.argv:
        gen_parametrizations()
        return
    if "--kl-only" in sys.argv:
        for tag in ("A_numeric", "B_ragged", "M_mixed"):
            gen_kl(tag)
        return
    if "--quantile-only" in sys.argv:
        gen_quantile()
        return
    if "--loader-only" in sys.argv:
        gen_loader()
        return
    if "--edm-only" in sys.argv:
        gen_edm()
        return
    if "--label-only" in sys.argv:
        gen_label_embedding()
        return
    if "--ddimx-only" in sys.argv:
        gen_ddim_family()
        return
    if "--ddim-only" in sys.argv:
        gen_ddim("A_numeric", [0], 8, [64, 64], 12, 48)
        gen_ddim("B_ragged", [3, 7, 10, 2, 12], 6, [128, 128], 16, 64)
        return
    if "--guided-only" in sys.argv:
        gen_guided("A_numeric", 8, [64, 64], 12, 48)
        return
    if "--classifier-only" in sys.argv:
        gen_classifier()
        return
    if "--berka-only" in sys.argv:
        gen_train_berka()
        return
    if "--train-only" not in sys.argv:
        main_cases()
        gen_guided("A_numeric", 8, [64, 64], 12, 48)
        gen_classifier()
        gen_loader()
        gen_quantile()
        gen_affine()
        for tag in ("A_numeric", "B_ragged", "M_mixed"):
            gen_kl(tag)
        gen_ddim("A_numeric", [0], 8, [64, 64], 12, 48)
        gen_ddim("B_ragged", [3, 7, 10, 2, 12], 6, [128, 128], 16, 64)
        gen_ddim_family()
        gen_edm()
        gen_label_embedding()
        gen_parametrizations()
    # training: 12 iterations of the reference trainer, mixed and numeric-only shapes
    gen_train("B_ragged", [3, 7, 10, 2, 12], 6, [64, 64], 1000, 24, 12, 2e-3, 1e-4)
    gen_train("A_numeric", [0], 8, [64, 64], 100, 24, 12, 2e-3, 0.0)
    gen_train_berka()


def main_cases():
    gen_schedule("cosine_1000", 1000, "cosine")
    gen_schedule("linear_100", 100, "linear")
    # A: the shipped pipeline's shape (numeric only, num_classes=[0]); also the Tartan-Federer shape
    gen_case("A_numeric", [0], 8, [64, 64], 100, 96, T_chain=12)
    # B: ragged categorical blocks incl. a 2-level one
    gen_case("B_ragged", [3, 7, 10, 2, 12], 6, [128, 128], 1000, 96, T_chain=16)
    # M: BASELINE config M (20 numeric + 10x10 levels), 4x512 denoiser
    gen_case("M_mixed", [10] * 10, 20, [512] * 4, 1000, 64, T_chain=8)
    # N: Tartan-Federer shape, 4x512
    gen_case("N_tf", [0], 20, [512] * 4, 1000, 64, P=4)
    # L: linear schedule, categorical only
    gen_case("L_catonly", [4, 4, 5], 0, [64, 64], 100, 48, sched="linear", T_chain=40)
    # K1: categorical columns with a single level (log K = 0; the posterior has no "other" column)
    gen_case("K1_single", [1, 3, 1], 2, [32, 32], 50, 32, T_chain=6)


if __name__ == "__main__":
    main()
