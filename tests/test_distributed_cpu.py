"""world_size-2 `gloo` tests of the N > 1 host logic: shard ranges, the gather of per-rank
score blocks, and the data-parallel gradient convention of mtk_train_step (each rank scales by 1 / global batch,
SUM all-reduce).  The per-rank compute is done by the CPU oracle here (no GPU in this tier); the GPU kernels
themselves are covered by tests/test_gpu_*.py, and tests/test_gpu_train.py::test_two_rank_training runs the same
flow on the device."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import diffusion_oracle as orc


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(fn, ws, *args):
    port = free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_entry, args=(fn, r, ws, port, q) + args) for r in range(ws)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    out = {}
    while not q.empty():
        r, v = q.get()
        out[r] = v
    return out


def _entry(fn, rank, ws, port, q, *args):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        q.put((rank, fn(rank, ws, *args)))
    finally:
        dist.destroy_process_group()


def test_shard_range_covers_everything():
    import midst_toolkit_amd.distributed as D

    for n in (0, 1, 7, 8, 100_001):
        for ws in (1, 2, 3, 8):
            blocks = [D.shard_range(n, r, ws) for r in range(ws)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            assert max(b[1] - b[0] for b in blocks) - min(b[1] - b[0] for b in blocks) <= 1


def _score_rank(rank, ws, n):
    import midst_toolkit_amd.distributed as D

    Nn, T = 6, 100
    s = orc.make_schedule(T, "cosine")
    params = orc.init_mlp_params(Nn, [32, 32], seed=0)
    g = torch.Generator().manual_seed(0)
    x = torch.randn((n, Nn), generator=g)
    noise = torch.randn((3, Nn), generator=g)
    lo, hi = D.shard_range(n, rank, ws)
    local = torch.stack([orc.tf_score(s, Nn, params, x[lo:hi], noise, t) for t in (5, 50)])  # [n_t, n_local, P]
    full = D.gather_rows(local, n, dim=1)
    want = torch.stack([orc.tf_score(s, Nn, params, x, noise, t) for t in (5, 50)])
    return float((full - want).abs().max()), tuple(full.shape)


@pytest.mark.parametrize("n", [64, 37])
def test_gather_of_sharded_scores(n):
    out = run_ranks(_score_rank, 2, n)
    for r in (0, 1):
        err, shape = out[r]
        assert shape == (2, n, 3)
        assert err < 1e-6, err   # each rank's rows are computed from the same inputs as the single-process run


def _grad_rank(rank, ws):
    import midst_toolkit_amd.distributed as D

    K, Nn, T, B = [3, 5], 4, 100, 16
    s = orc.make_schedule(T, "cosine")
    params = orc.init_mlp_params(Nn + sum(K), [32, 32], seed=0)
    g = torch.Generator().manual_seed(1)
    x0 = torch.cat([torch.randn((B, Nn), generator=g)] + [torch.randint(0, k, (B, 1), generator=g).float() for k in K], 1)
    t = torch.randint(0, T, (B,), generator=g)
    eps, u = torch.randn((B, Nn), generator=g), torch.rand((B, sum(K)), generator=g)
    lo, hi = D.shard_range(B, rank, ws)
    # per-rank gradient of (sum over local rows) / global batch == local mean * (local rows / global rows)
    grads, _, _ = orc.loss_and_grads(s, K, Nn, params, x0[lo:hi], t[lo:hi], eps[lo:hi], u[lo:hi])
    flat = torch.cat([v.reshape(-1) for v in grads.values()]) * ((hi - lo) / B)
    D.allreduce_gradients(flat)
    want, _, _ = orc.loss_and_grads(s, K, Nn, params, x0, t, eps, u)
    wflat = torch.cat([v.reshape(-1) for v in want.values()])
    return float((flat - wflat).norm() / wflat.norm())


def test_data_parallel_gradient_is_the_global_batch_gradient():
    out = run_ranks(_grad_rank, 2)
    assert out[0] < 1e-5 and out[1] < 1e-5, out
