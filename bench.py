#!/usr/bin/env python3
"""bench.py — record x timestep denoising-loss scoring throughput on MI355X.

Workload (BASELINE.json configs[1]): Tartan-Federer-style scoring, 100k records per GPU x timesteps,
4x512 MLPDiffusion, 20 numeric features (shape N of SURVEY.md section 8), T = 1000, P = 1 noise draw.
One "step" = one mtk_tf_score call over all resident records and `--nt` timesteps
(noising -> denoiser -> per-record MSE).  Inputs are resident in HBM before the timed region.
With N > 1 GPUs the records are sharded (weak scaling: 100k per GPU) and each step ends with the RCCL
all-gather of the per-record score vectors.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the denoiser MFMA kernel);
`cpu_baseline` times the CPU oracle (a port of the reference's op sequence in PyTorch-CPU eager) on a
bounded sample on this box's host cores.
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

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}  # dense MFMA peaks, MI355X_MICROARCH.md


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


class _HostGather:
    """Rehearsal only (MTK_BENCH_BACKEND=gloo): an asynchronous all-gather through host tensors with the
    .wait() of a torch.distributed work handle."""

    def __init__(self, dist, src, dst, world):
        self.host = [torch.empty(src.shape, dtype=src.dtype) for _ in range(world)]
        self.dst = dst
        self.work = dist.all_gather(self.host, src.cpu(), async_op=True)

    def wait(self):
        self.work.wait()
        self.dst.copy_(torch.stack(self.host))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--records", type=int, default=100_000, help="records per GPU")
    ap.add_argument("--nt", type=int, default=10, help="timesteps per step")
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    if os.environ.get("MTK_BENCH_CHECK"):  # rehearsal runs: a hang ends with every thread's stack on stderr
        import faulthandler

        faulthandler.dump_traceback_later(60, exit=True)
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
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import midst_toolkit_amd as mtk
    from midst_toolkit_amd import scoring

    T, Nn, P, seed = 1000, 20, 1, 20250821
    hidden = [a.hidden] * 4
    torch.manual_seed(0)
    net = mtk.MLPDiffusion(Nn, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters(hidden, 0.0))
    params = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net.precision = a.precision
    diff = mtk.GaussianMultinomialDiffusion(np.array([0]), Nn, net, num_timesteps=T, device=dev).to(dev)
    diff.precision = a.precision

    g = torch.Generator().manual_seed(seed + rank)
    x = torch.randn((a.records, Nn), generator=g).clamp(-5.2, 5.2).to(dev)  # this rank's shard, resident
    noise = torch.randn((P, Nn), generator=torch.Generator().manual_seed(seed)).to(dev)  # shared table
    eng = diff.engine()
    # N > 1: the only exchange of the path is the gather of the score blocks.  It runs on RCCL's own stream while the
    # next step's kernel computes into the other buffer pair (double buffered); a step waits, stream-side, only for
    # the gather that last used its pair.  Every gather is drained inside the timed region.
    nbuf = 2 if world > 1 else 1
    scores = [torch.empty((a.nt, a.records, P), dtype=torch.float32, device=dev) for _ in range(nbuf)]
    gathered = [torch.empty((world, a.nt, a.records, P), dtype=torch.float32, device=dev) for _ in range(nbuf)] if world > 1 else None
    pending = [None] * nbuf

    def step(i, events=None):
        b = i % nbuf
        if pending[b] is not None:
            pending[b].wait()
            pending[b] = None
        ts = [(i * a.nt + k) % T for k in range(a.nt)]
        if events:
            events[0].record()
        eng.tf_score(x, noise, ts, 0, a.precision, out=scores[b])
        if events:
            events[1].record()
        if world > 1:
            if backend == "nccl":
                pending[b] = dist.all_gather_into_tensor(gathered[b], scores[b], async_op=True)
            else:  # gloo rehearsal of the control flow: staged through host tensors
                pending[b] = _HostGather(dist, scores[b], gathered[b], world)

    def barrier():
        for b in range(nbuf):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
        torch.cuda.synchronize()
        if world > 1:
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
    if world > 1 and os.environ.get("MTK_BENCH_CHECK"):  # rehearsal: every rank's block arrived in rank order
        last = (a.warmup + a.steps - 1) % nbuf
        assert torch.equal(gathered[last][rank], scores[last])
        mine = torch.stack([t.flatten()[:4] for t in gathered[last]])
        ref = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(ref, mine)
        assert all(torch.equal(r, mine) for r in ref)
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    kern_ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))  # HIP events on the launch stream

    # HBM-side bytes per launch of the dominant kernel, from the committed PMC passes of this same configuration
    traffic = None
    pmc = os.path.join(ROOT, "benchmarks", "pmc_fused.json")  # copy of profiles/r1_pmc_fused.json that travels to the GPU box
    if a.precision == "bf16" and a.records == 100_000 and a.nt == 10 and a.hidden == 512 and os.path.exists(pmc):
        with open(pmc) as f:
            rec = json.load(f)
        traffic = rec["fetch_bytes_per_launch_corrected_x2"] + rec["write_bytes_per_launch"]
    evals_step = a.records * a.nt * P
    fl = flops_per_eval(Nn, 128, hidden)
    achieved = evals_step * fl / (kern_ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[a.precision]
    out = {
        "metric": "record·timestep loss evals/sec", "value": world * evals_step * a.steps / dt, "unit": "evals/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if a.precision == "bf16" else "f32", "data": "synthetic",
        "config": {"workload": f"Tartan-Federer denoising-loss scoring (BASELINE configs[1]): {a.records} records/GPU x {a.nt} timesteps/step, "
                               f"P=1, 20 numeric features, 4x{a.hidden} MLPDiffusion, T=1000; random-init weights",
                   "records_per_gpu": a.records, "timesteps_per_step": a.nt, "hidden": a.hidden, "sharding": f"records/{world}"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                     "kernel": "fused tf-score kernel (noise + denoiser + MSE)" if a.precision == "bf16" else "layered f32-MFMA denoiser (whole step)",
                     "flops_per_eval": fl, "kernel_ms": kern_ms},
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(params, Nn, T, seed)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
