#!/usr/bin/env python3
"""bench.py — record x timestep denoising-loss scoring throughput (+ sampled rows/s) on MI355X.

Headline workload (BASELINE.json configs[1]): Tartan-Federer-style scoring of 100k records per GPU x ALL 1000 timesteps,
4x512 MLPDiffusion, 20 numeric features (shape N of SURVEY.md section 8), P = 1 noise draw.  One "step" = one
mtk_tf_score call over all resident records and all `--nt` timesteps (noising -> denoiser -> per-record MSE, 1e8 evals,
~150 ms).  Inputs are resident in HBM before the timed region.

N > 1 GPUs (one process per GPU, RCCL): `--mode weak` (default) keeps 100k records per GPU; `--mode strong` is
BASELINE configs[2]: 1M records x P = 4 in total, sharded with distributed.shard_range.  Either way the only exchange is
the gather of each rank's score block to the consumer rank 0, asynchronous and double buffered, drained inside the
timed region.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel of the headline workload (the fused
noise + denoiser + MSE MFMA kernel); `cpu_baseline` times the CPU oracle (a port of the reference's op sequence in
PyTorch-CPU eager) on a bounded sample on this box's host cores; `extra` carries the other half of the metric and the
other modes, each with its own roofline fraction, measured after the timed region (N = 1 only unless noted):
  sampling_4x512 / sampling_4x1024  sampled rows/s at T = 1000, shape M (20 numeric + 10 x 10 categorical), every rank
  mixed_loss_shapeM                 evals/s of q_sample -> denoiser -> loss_rows at shape M with the three kernels' rates
  parity_fp32 / parity_bf16x3       the headline workload in the two modes that carry the 1e-4 parity bar
  training_2048x4x1024              one ClavaDDPMTrainer iteration at BASELINE configs[4]'s per-rank shape, bf16 and fp32 GEMMs
  sampling_guided_caller_sized      microseconds per classifier-guided reverse step at 1,024 / 8,192 rows (ClavaDDPM default denoiser and classifier)
  sampling_caller_sized             microseconds per reverse step at 1,024 / 8,192 rows (what the reference's callers sample per call):
                                    4 x 512, 4 x 1024 and ClavaDDPM's default 512-1024-1024-1024-1024-512 denoiser
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "bf16x3": 2500.0, "fp32": 157.3}  # dense MFMA peaks, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
DTYPE = {"bf16": "bf16", "bf16x3": "bf16 (hi+lo split operands, 3 MFMAs per product, fp32 accumulate)", "fp32": "f32"}
TF_LAUNCH = 64  # timesteps per launch of the fused Tartan-Federer kernel (kMaxTF, mtk_mlp_bf16.hip)


def flops_per_eval(D: int, dim_t: int, hidden: list[int]) -> int:
    dims = [D, dim_t] + list(hidden) + [D]
    return 2 * sum(a * b for a, b in zip(dims[:-1], dims[1:]))


def cpu_baseline(params, Nn, T, seed, seconds_target=10.0):
    """The CPU oracle on a bounded sample: 8192 records x k timesteps, scaled to ~10 s of work."""
    from oracle import diffusion_oracle as orc

    s = orc.make_schedule(T, "cosine")
    g = torch.Generator().manual_seed(seed)
    n_rec = 8192
    x = torch.randn((n_rec, Nn), generator=g).clamp(-5.2, 5.2)
    noise = torch.randn((1, Nn), generator=g)
    with torch.no_grad():
        orc.tf_score(s, Nn, params, x[:512], noise, 3)  # warm-up
        t0 = time.perf_counter()
        n_t = 0
        while time.perf_counter() - t0 < seconds_target and n_t < 2000:
            orc.tf_score(s, Nn, params, x, noise, (37 * n_t + 1) % T)
            n_t += 1
        dt = time.perf_counter() - t0
    return {"value": n_rec * n_t / dt, "unit": "evals/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_rec} records x {n_t} timesteps x 1 noise draw, oracle/diffusion_oracle.tf_score (PyTorch-CPU eager), {dt:.1f} s"}


def cpu_baseline_sampling(seed, seconds_target=6.0):
    """The CPU oracle's reverse chain (shape M, 4x512) on a bounded sample: 1024 rows x k steps, scaled to row-steps/s."""
    from oracle import diffusion_oracle as orc

    K, Nn, T = np.array([10] * 10), 20, 1000
    D = Nn + int(K.sum())
    s = orc.make_schedule(T, "cosine")
    params = orc.init_mlp_params(D, [512] * 4, seed=0)
    g = torch.Generator().manual_seed(seed)
    B = 1024
    z = torch.randn((B, Nn), generator=g, dtype=torch.float64)
    zi = torch.stack([torch.randint(0, int(k), (B,), generator=g) for k in K], 1)
    steps = 0
    with torch.no_grad():
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds_target and steps < T:
            t = T - 1 - steps
            tv = torch.full((B,), t, dtype=torch.long)
            out = orc.mlp_forward(params, orc.model_input(z.float(), zi, K), tv)
            z = orc.gaussian_p_step(s, out[:, :Nn], z, t, torch.randn((B, Nn), generator=g, dtype=torch.float64))
            logp = orc.q_posterior(s, orc.block_log_softmax(out[:, Nn:], K), orc.log_onehot(zi, K), tv, K)
            zi = orc.gumbel_argmax(logp, torch.rand((B, int(K.sum())), generator=g), K)
            steps += 1
        dt = time.perf_counter() - t0
    rs = B * steps / dt
    return {"value": rs / T, "unit": "sampled rows/s at T=1000", "row_steps_per_s": rs, "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{B} rows x {steps} reverse steps (denoiser + Gaussian posterior + multinomial posterior + Gumbel), oracle, {dt:.1f} s"}


class _HostGather:
    """Rehearsal only (MTK_BENCH_BACKEND=gloo): an asynchronous gather to rank 0 through host tensors with the
    .wait() of a torch.distributed work handle."""

    def __init__(self, dist, src, dst, world, rank):
        self.host = [torch.empty(src.shape, dtype=src.dtype) for _ in range(world)] if rank == 0 else None
        self.dst, self.rank = dst, rank
        self.work = dist.gather(src.cpu(), self.host, dst=0, async_op=True)

    def wait(self):
        self.work.wait()
        if self.rank == 0:
            for d, h in zip(self.dst, self.host):
                d.copy_(h)


def timeit(fn, iters=5, warm=2):
    """Median HIP-event time (s) of fn on torch's current stream (the stream every libmtk_hip launch goes to)."""
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


def bench_sampling(mtk, dev, hidden, rows, dist, world, comm=False, warm_rows=None):
    """Sampled rows/s at T = 1000, shape M, in-kernel Philox noise (math "fast"), bf16 denoiser: every rank runs its own
    chain over `rows` rows (no collective inside the loop); value = world x rows / max-over-ranks time."""
    K, Nn, T = np.array([10] * 10), 20, 1000
    D = Nn + int(K.sum())
    torch.manual_seed(0)
    net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([hidden] * 4, 0.0))
    net.precision = "bf16"
    d = mtk.GaussianMultinomialDiffusion(K, Nn, net, num_timesteps=T, device=dev).to(dev)
    d.precision, d.rng, d.results_on_device = "bf16", "philox", True
    d.sample(warm_rows or rows, torch.ones(1), seed=1)  # warm-up: packs the weights, sizes the workspaces
    if warm_rows and warm_rows < rows:  # a small warm-up chain, then the workspaces and the first-touch of the timed batch's buffers
        d.engine().reserve(rows)
    torch.cuda.synchronize()
    if comm:
        dist.barrier()
    t0 = time.perf_counter()
    x, _ = d.sample(rows, torch.ones(1), seed=2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if comm:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    assert x.shape == (rows, Nn + len(K)) and bool(torch.isfinite(x).all())
    # the two kernels of one reverse step, timed alone at the chain's batch size
    eng = d.engine()
    x_in = torch.randn((D, rows), device=dev)
    out = torch.empty((D, rows), device=dev)
    z = torch.randn((Nn, rows), device=dev, dtype=torch.float64)
    zi = torch.randint(0, 10, (len(K), rows), device=dev, dtype=torch.int32)
    t_mlp = timeit(lambda: eng.mlp_forward(x_in, 500, out=out, precision="bf16", x_strides=(1, rows), out_strides=(1, rows), B=rows))
    t_step = timeit(lambda: eng.p_sample_step(rows, 500, False, out, (1, rows), z, zi, None, None, x_in, (1, rows), seed=3))
    fl = flops_per_eval(D, 128, [hidden] * 4)
    b_step = Nn * 8 * 2 + Nn * 4 + int(K.sum()) * 4 + len(K) * 4 * 2 + D * 4  # 1,360 B per row-step (Philox)
    return {"sampled_rows_per_s": world * rows / dt, "row_steps_per_s": world * rows * T / dt, "rows_per_gpu": rows, "T": T, "seconds": dt,
            "denoiser": f"4x{hidden} bf16", "rng": "philox (in-kernel)", "math": "fast", "launches": 2 * T + 2,
            "roofline_denoiser": {"bound": "mfma", "achieved": rows * fl / t_mlp / 1e12, "peak": PEAK_TFLOPS["bf16"], "unit": "TFLOP/s",
                                  "frac": rows * fl / t_mlp / 1e12 / PEAK_TFLOPS["bf16"], "kernel_ms": t_mlp * 1e3, "flops_per_row_step": fl},
            "roofline_step": {"bound": "hbm", "achieved": rows * b_step / t_step / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": rows * b_step / t_step / 1e9 / HBM_PEAK_GBS, "kernel_ms": t_step * 1e3, "bytes_per_row_step": b_step},
            "whole_loop_mfma_frac": rows * T * fl / dt / 1e12 / PEAK_TFLOPS["bf16"]}


def bench_sampling_small(mtk, dev):
    """Reverse chains at the batch sizes the reference's callers sample per call (SYNTH:79: 1 k - 20 k rows): microseconds per
    reverse step (denoiser + posterior step) at T = 1000, shape M, in-kernel Philox noise, bf16 denoiser."""
    K, Nn, T = np.array([10] * 10), 20, 1000
    D = Nn + int(K.sum())
    out = {"T": T, "rng": "philox (in-kernel)", "math": "fast", "unit": "us per reverse step"}
    # 4 x 512 (BASELINE configs[1]'s denoiser), 4 x 1024 (configs[3] / [4]) and ClavaDDPM's default six-layer denoiser
    for name, layers in (("4x512", [512] * 4), ("4x1024", [1024] * 4), ("clavaddpm_default_512_1024x4_512", [512, 1024, 1024, 1024, 1024, 512])):
        torch.manual_seed(0)
        net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters(list(layers), 0.0))
        net.precision = "bf16"
        d = mtk.GaussianMultinomialDiffusion(K, Nn, net, num_timesteps=T, device=dev).to(dev)
        d.precision, d.rng, d.results_on_device = "bf16", "philox", True
        for rows in (1024, 8192):
            d.sample(rows, torch.ones(1), seed=1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            x, _ = d.sample(rows, torch.ones(1), seed=2)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            assert bool(torch.isfinite(x).all())
            out[f"{name}_{rows}_rows"] = {"us_per_step": dt / T * 1e6, "sampled_rows_per_s": rows / dt}
    return out


def bench_sampling_guided(mtk, dev):
    """Classifier-guided reverse chain (GMD:1319 conditional_sample, what ClavaDDPM runs for child tables) at caller-sized batches:
    microseconds per step with ClavaDDPM's default denoiser (bf16) and default guidance classifier (fp32: parity-grade gradient)."""
    Nn, T, n_cls = 20, 200, 12
    torch.manual_seed(0)
    net = mtk.MLPDiffusion(Nn, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([512, 1024, 1024, 1024, 1024, 512], 0.0))
    net.precision = "bf16"
    d = mtk.GaussianMultinomialDiffusion(np.asarray([0]), Nn, net, num_timesteps=T, device=dev).to(dev)
    d.precision, d.rng, d.results_on_device = "bf16", "philox", True
    clf = mtk.Classifier(Nn, n_cls, 128, [128, 256, 512, 1024, 512, 256, 128], dropout_prob=0.1).to(dev).eval()
    cond = mtk.get_conditioning_function(clf, 1.0)
    out = {"T": T, "denoiser": "512-1024x4-512 bf16", "classifier": "128-256-512-1024-512-256-128 fp32", "numeric_features": Nn, "unit": "us per guided reverse step"}
    for rows in (1024, 8192):
        y = torch.randint(0, n_cls, (rows,), device=dev)
        d.conditional_sample(y.cpu(), model_kwargs={"y": y}, conditioning_function=cond)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x, _ = d.conditional_sample(y.cpu(), model_kwargs={"y": y}, conditioning_function=cond)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert bool(torch.isfinite(x).all())
        out[f"{rows}_rows"] = {"us_per_step": dt / T * 1e6, "sampled_rows_per_s_at_T1000": rows / (dt / T * 1000)}
    return out


def bench_mixed_loss(mtk, dev, rows):
    """Shape-M mixed_loss (q_sample -> fused bf16 denoiser -> loss_rows), in-kernel Philox noise, math "fast"."""
    K, Nn, T = np.array([10] * 10), 20, 1000
    C, S = len(K), int(K.sum())
    D = Nn + S
    torch.manual_seed(0)
    net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([512] * 4, 0.0))
    net.precision = "bf16"
    d = mtk.GaussianMultinomialDiffusion(K, Nn, net, num_timesteps=T, device=dev).to(dev)
    d.precision, d.rng = "bf16", "philox"
    eng = d.engine()
    g = torch.Generator().manual_seed(20250821)
    x0 = torch.cat([torch.randn((rows, Nn), generator=g).clamp(-5.2, 5.2)] + [torch.randint(0, int(k), (rows, 1), generator=g).float() for k in K], 1).to(dev)
    t = torch.randint(0, T, (rows,), generator=g).to(dev)
    x_in = torch.empty((D, rows), device=dev)
    out = torch.empty((D, rows), device=dev)
    idx = torch.empty((C, rows), dtype=torch.int32, device=dev)
    lm, lg = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    q = lambda: eng.q_sample(x0, t, None, None, x_in, (1, rows), idx, seed=1)  # noqa: E731
    m = lambda: eng.mlp_forward(x_in, t, out=out, precision="bf16", x_strides=(1, rows), out_strides=(1, rows), B=rows)  # noqa: E731
    lo = lambda: eng.loss_rows(out, (1, rows), x0, idx, None, t, lm, lg, seed=1)  # noqa: E731
    tq, tm, tl = timeit(q), timeit(m), timeit(lo)
    tot = timeit(lambda: d.mixed_loss_rows(x0, timestep=t, seed=7))
    fl = flops_per_eval(D, 128, [512] * 4)
    b_q = (Nn + C) * 4 + 8 + D * 4 + C * 4          # 648 B per row (Philox)
    b_l = D * 4 + (Nn + C) * 4 + C * 4 + 8 + 8      # 656 B per row
    hbm = lambda b, s: {"bound": "hbm", "achieved": rows * b / s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rows * b / s / 1e9 / HBM_PEAK_GBS,  # noqa: E731
                        "kernel_ms": s * 1e3, "bytes_per_row": b}
    return {"evals_per_s": rows / tot, "rows": rows, "ms": tot * 1e3, "denoiser": "4x512 bf16 (fused)", "rng": "philox (in-kernel)", "math": "fast",
            "roofline_q_sample": hbm(b_q, tq), "roofline_loss_rows": hbm(b_l, tl),
            "roofline_denoiser": {"bound": "mfma", "achieved": rows * fl / tm / 1e12, "peak": PEAK_TFLOPS["bf16"], "unit": "TFLOP/s",
                                  "frac": rows * fl / tm / 1e12 / PEAK_TFLOPS["bf16"], "kernel_ms": tm * 1e3, "flops_per_eval": fl},
            "note": "each kernel's time includes the 8 B/row pass that turns rows with an out-of-range timestep into NaN"}


def bench_training(mtk, dev, hidden=1024, batch=2048, steps=200):
    """BASELINE configs[4]'s per-rank shape (global batch 16,384 over 8 ranks = 2,048 rows, 4 x 1024, shape M): one
    ClavaDDPMTrainer iteration = noising + forward + losses + backward + AdamW + EMA, in-kernel Philox noise and timesteps."""
    K, Nn, T = np.array([10] * 10), 20, 1000
    D = Nn + int(K.sum())
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1)
    x0 = torch.cat([torch.randn((batch, Nn), generator=g)] + [torch.randint(0, int(k), (batch, 1), generator=g).float() for k in K], 1).to(dev)
    fl_fwd = flops_per_eval(D, 128, [hidden] * 4)
    fl_step = batch * (3 * fl_fwd - 2 * D * 128)  # forward + data gradients + weight gradients; the input has no gradient
    out = {"rows": batch, "denoiser": f"4x{hidden}", "rng": "philox (in-kernel, timesteps included)", "reductions": "deterministic",
           "includes": "q_sample, forward, losses, backward, AdamW, EMA; no host-framework kernel in the iteration"}
    for prec in ("bf16", "fp32"):
        net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([hidden] * 4, 0.0))
        d = mtk.GaussianMultinomialDiffusion(K, Nn, net, num_timesteps=T, device=dev).to(dev)
        d.rng = "philox"
        tr = mtk.ClavaDDPMTrainer(d, iter(()), 2e-4, 1e-4, steps, device=dev)
        tr.set_train_precision(prec)
        for _ in range(6):
            tr._train_step(x0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr._train_step(x0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        peak = PEAK_TFLOPS["bf16" if prec == "bf16" else "fp32"]
        out[prec] = {"ms_per_step": dt * 1e3, "rows_per_s": batch / dt,
                     "roofline": {"bound": "mfma", "achieved": fl_step / dt / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": fl_step / dt / 1e12 / peak,
                                  "note": "whole iteration (32 launches at 2,048 rows), not one kernel"}}
        tr.finish()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--records", type=int, default=0, help="records per GPU (weak; default 100000) or in total (strong; default 1000000)")
    ap.add_argument("--nt", type=int, default=0, help="timesteps per step (default: all 1000 in weak mode, 100 in strong mode)")
    ap.add_argument("--P", type=int, default=0, help="noise draws per record (default 1 weak, 4 strong)")
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16x3", "fp32"])
    ap.add_argument("--mode", default="weak", choices=["weak", "strong"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--sample-rows", type=int, default=262144)
    ap.add_argument("--sample-rows-large", type=int, default=1 << 20, help="second sampling figure: one chain of BASELINE configs[3]'s 1M rows (0: skip)")
    ap.add_argument("--rehearse-rccl", action="store_true",
                    help="one GPU: initialise an RCCL (nccl) process group of ONE rank and run the N > 1 code path on it - the asynchronous, "
                         "double-buffered gather of the score blocks, the barriers, the MAX all-reduce of the time (what a one-GPU box can "
                         "show of SURVEY 8e)")
    a = ap.parse_args()

    if os.environ.get("MTK_BENCH_CHECK"):  # rehearsal runs: a hang ends with every thread's stack on stderr
        import faulthandler

        faulthandler.dump_traceback_later(120, exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with {a.gpus} ranks (WORLD_SIZE={world})")
    # MTK_BENCH_BACKEND=gloo rehearses the N > 1 control flow with several ranks on one card (RCCL refuses that)
    backend = os.environ.get("MTK_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    comm = world > 1 or a.rehearse_rccl  # the collectives of the N > 1 path run (at N = 1: the rehearsal)
    if comm:
        import torch.distributed as dist

        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import midst_toolkit_amd as mtk
    from midst_toolkit_amd.distributed import shard_range

    strong = a.mode == "strong"
    T, Nn, seed = 1000, 20, 20250821
    P = a.P or (4 if strong else 1)
    nt = a.nt or (100 if strong else T)
    total = a.records or (1_000_000 if strong else 100_000)
    if strong:
        lo, hi = shard_range(total, rank, world)
        n_local, n_max = hi - lo, shard_range(total, 0, world)[1]
        n_global = total
    else:
        lo, n_local, n_max, n_global = rank * total, total, total, total * world
    hidden = [a.hidden] * 4
    torch.manual_seed(0)
    net = mtk.MLPDiffusion(Nn, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters(hidden, 0.0))
    params = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net.precision = a.precision
    diff = mtk.GaussianMultinomialDiffusion(np.array([0]), Nn, net, num_timesteps=T, device=dev).to(dev)
    diff.precision = a.precision

    # this rank's shard, resident; rows are a pure function of their global index (strong mode: any world size scores
    # the same 1M records)
    g = torch.Generator().manual_seed(seed)
    if strong:
        x = torch.randn((total, Nn), generator=g).clamp(-5.2, 5.2)[lo:lo + n_local].contiguous().to(dev)
    else:
        x = torch.randn((n_local, Nn), generator=torch.Generator().manual_seed(seed + rank)).clamp(-5.2, 5.2).to(dev)
    noise = torch.randn((P, Nn), generator=torch.Generator().manual_seed(seed)).to(dev)  # shared table
    eng = diff.engine()
    # N > 1: the only exchange of the path is the gather of the score blocks to the consumer (rank 0).  It runs on RCCL's
    # own stream while the next step's kernels compute into the other buffer (double buffered); a step waits,
    # stream-side, only for the gather that last used its buffer.  Every gather is drained inside the timed region.
    nbuf = 2 if comm else 1
    scores = [torch.empty((nt, n_max, P), dtype=torch.float32, device=dev) for _ in range(nbuf)]  # padded to the largest shard
    gathered = [[torch.empty((nt, n_max, P), dtype=torch.float32, device=dev) for _ in range(world)] for _ in range(nbuf)] if (comm and rank == 0) else None
    pending = [None] * nbuf

    ragged = n_local != n_max  # strong scaling with total % world != 0: score into a compact block, copy into the padded buffer
    compact = torch.empty((nt, n_local, P), dtype=torch.float32, device=dev) if ragged else None

    def step(i, events=None):
        b = i % nbuf
        if pending[b] is not None:
            pending[b].wait()
            pending[b] = None
        ts = [(i * nt + k) % T for k in range(nt)]
        if events:
            events[0].record()
        eng.tf_score(x, noise, ts, 0, a.precision, out=compact if ragged else scores[b])
        if ragged:
            scores[b][:, :n_local].copy_(compact)
        if events:
            events[1].record()
        if comm:
            if backend == "nccl":
                pending[b] = dist.gather(scores[b], gathered[b] if rank == 0 else None, dst=0, async_op=True)
            else:  # gloo rehearsal of the control flow: staged through host tensors
                pending[b] = _HostGather(dist, scores[b], gathered[b] if rank == 0 else None, world, rank)

    def barrier():
        for b in range(nbuf):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
        torch.cuda.synchronize()
        if comm:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i, ev[i])
    barrier()
    dt = time.perf_counter() - t0
    if comm and (os.environ.get("MTK_BENCH_CHECK") or a.rehearse_rccl):  # rehearsal: every rank's block arrived at the consumer, in rank order
        last = (a.warmup + a.steps - 1) % nbuf
        mine = scores[last].flatten()[:4].cpu()
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine) if backend != "nccl" else dist.all_gather(every := [t.to(dev) for t in every], mine.to(dev))
        if rank == 0:
            assert torch.equal(gathered[last][0], scores[last])
            for r in range(world):
                assert torch.equal(gathered[last][r].flatten()[:4].cpu(), every[r].cpu()), r
    if comm:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    step_ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))  # HIP events on the launch stream, around one mtk_tf_score call

    fused = a.precision == "bf16" and a.hidden == 512
    launches = (nt + TF_LAUNCH - 1) // TF_LAUNCH if fused else None
    kern_ms = step_ms / launches if fused else step_ms
    # HBM-side bytes per launch of the dominant kernel, from the committed PMC passes of this same configuration
    traffic = None
    pmc = os.path.join(ROOT, "benchmarks", "pmc_fused.json")  # copy of profiles/r2_pmc_fused.json that travels to the GPU box
    if fused and n_local == 100_000 and nt == 1000 and P == 1 and os.path.exists(pmc):
        with open(pmc) as f:
            rec = json.load(f)
        if rec.get("timesteps_per_step") == nt:
            traffic = rec["fetch_bytes_per_launch_corrected_x2"] + rec["write_bytes_per_launch"]
    evals_step = n_local * nt * P
    fl = flops_per_eval(Nn, 128, hidden)
    achieved = evals_step * fl / (step_ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[a.precision]
    # MFMA FLOPs the fused kernel actually issues per eval: proj is folded into the first hidden layer (same function)
    # and every operand is padded to MFMA tiles: (16 tiles x 2 DT + 3 x 16 x 32 + DT x 32) MFMAs of 32x32x16 per 32 records
    dt_tiles = (Nn + 31) // 32
    executed = (16 * 2 * dt_tiles + 3 * 16 * 32 + dt_tiles * 32) * 1024 if (fused and a.hidden == 512) else None
    kernel_name = {"bf16": "k_fused<512,1,1>: fused Tartan-Federer kernel (noising + denoiser + MSE), one launch per 64 timesteps" if fused
                   else "layered bf16 MFMA denoiser between noising and MSE kernels (whole step)",
                   "bf16x3": "layered split-bf16 MFMA denoiser (k_gemm_bf16<SPLIT>) between noising and MSE kernels (whole step)",
                   "fp32": "layered f32-MFMA denoiser (k_gemm_f32) between noising and MSE kernels (whole step)"}[a.precision]
    if strong:
        workload = (f"Loss-based MIA scoring (BASELINE configs[2]): {total} records x {nt} timesteps/step x P={P} noise draws in total, sharded over "
                    f"{world} GPU(s) with shard_range, score blocks gathered to rank 0")
    else:
        workload = (f"Tartan-Federer denoising-loss scoring (BASELINE configs[1]): {total} records/GPU x {nt} timesteps/step (T=1000: "
                    f"{'every timestep' if nt == T else 'a slice'}), P={P}")
    workload += (f", 20 numeric features, 4x{a.hidden} MLPDiffusion, precision {a.precision}; injected shared [P,Nn] noise table (TF semantics: no "
                 "in-kernel RNG on this path), math: reference formulas (noising is one fp32 FMA pair); random-init weights")
    out = {
        "metric": "record·timestep loss evals/sec (node) + sampled rows/sec, T=1000", "value": (n_global if strong else world * n_local) * nt * P * a.steps / dt,
        "unit": "evals/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": DTYPE[a.precision], "data": "synthetic",
        "config": {"workload": workload, "records_per_gpu": n_local, "timesteps_per_step": nt, "noise_draws": P, "hidden": a.hidden,
                   "sharding": f"records/{world}", "evals_per_step": (n_global if strong else world * n_local) * nt * P},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                     "kernel": kernel_name, "flops_per_eval": fl, "kernel_ms": kern_ms, "launches_per_step": launches,
                     "flops_per_launch": evals_step * fl / launches if fused else evals_step * fl,
                     # `achieved` / `frac` use the ALGORITHMIC FLOPs of the reference's network; the kernel evaluates the same
                     # function with proj folded into the first hidden layer, so its matrix pipe is busy for this fraction of the peak:
                     "executed_mfma_flops_per_eval": executed,
                     "mfma_pipe_frac": (achieved / peak * executed / fl) if executed else None},
    }

    extras = {}
    if not a.no_extras and not strong:
        # the other half of the metric, on every rank (independent chains, no collective in the loop)
        extras["sampling_4x512"] = bench_sampling(mtk, dev, 512, a.sample_rows, dist, world, comm)
        extras["sampling_4x1024"] = bench_sampling(mtk, dev, 1024, a.sample_rows, dist, world, comm)
        out["sampled_rows_per_s"] = extras["sampling_4x1024"]["sampled_rows_per_s"]  # BASELINE configs[3]: 4x1024, shape M, T = 1000
        if a.sample_rows_large > a.sample_rows:  # configs[3] at its own size: ONE chain of 1M rows (beyond what the Infinity Cache holds)
            extras["sampling_4x1024_1M"] = bench_sampling(mtk, dev, 1024, a.sample_rows_large, dist, world, comm, warm_rows=8192)
            extras["sampling_4x512_1M"] = bench_sampling(mtk, dev, 512, a.sample_rows_large, dist, world, comm, warm_rows=8192)
            out["sampled_rows_per_s_1M_rows"] = extras["sampling_4x1024_1M"]["sampled_rows_per_s"]
        out["sampled_rows_per_s_4x512"] = extras["sampling_4x512"]["sampled_rows_per_s"]
    if not a.no_extras and world == 1 and not strong:
        extras["mixed_loss_shapeM"] = bench_mixed_loss(mtk, dev, 1 << 20)
        extras["training_2048x4x1024"] = bench_training(mtk, dev)
        extras["sampling_caller_sized"] = bench_sampling_small(mtk, dev)
        extras["sampling_guided_caller_sized"] = bench_sampling_guided(mtk, dev)
        for prec, nt_p in (("fp32", 10), ("bf16x3", 20)):  # the modes that carry the <= 1e-4 parity bar, same records
            if prec == a.precision:
                continue
            ts = list(range(0, T, T // nt_p))[:nt_p]
            buf = torch.empty((nt_p, n_local, P), dtype=torch.float32, device=dev)
            secs = timeit(lambda: eng.tf_score(x, noise, ts, 0, prec, out=buf), iters=3, warm=1)
            tf = n_local * nt_p * P * fl / secs / 1e12
            extras[f"parity_{prec}"] = {"evals_per_s": n_local * nt_p * P / secs, "records": n_local, "timesteps": nt_p, "ms": secs * 1e3,
                                        "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_TFLOPS[prec], "unit": "TFLOP/s",
                                                     "frac": tf / PEAK_TFLOPS[prec], "note": "bf16x3 issues 3 MFMAs per product: frac counts algorithmic FLOPs"},
                                        "parity": "<= 1e-4 relative vs the reference (tests/test_gpu_parity.py)"}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(params, Nn, T, seed)
        if extras:
            extras["cpu_baseline_sampling"] = cpu_baseline_sampling(seed)
    if extras:
        out["extra"] = extras
    if rank == 0:
        print(json.dumps(out), flush=True)
    if comm:
        if rank == 0 and a.rehearse_rccl:
            out_note = {"rehearse_rccl": "ok: nccl process group of %d rank(s); async double-buffered gather, barriers and MAX all-reduce ran" % world}
            print(json.dumps(out_note), file=sys.stderr, flush=True)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
