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


def _bucket_rank(rank, ws):
    import midst_toolkit_amd.distributed as D

    g = torch.Generator().manual_seed(100 + rank)
    image = torch.randn((40_000,), generator=g)
    layout = [(33_000, 7_000), (20_000, 13_000), (4_096, 15_904), (0, 4_096)]  # completion order: head first, proj side last
    blocking = image.clone()
    D.allreduce_gradients(blocking)
    ex = D.BucketedGradientExchange(None, image, layout=layout)
    assert not ex.overlapped and ex.world == ws
    ex.launch()
    ex.wait()
    return bool(torch.equal(image, blocking))


def test_bucketed_gradient_exchange_equals_the_blocking_all_reduce():
    """the per-bucket exchange (what the trainer issues under the backward on the GPU; here its blocking
    form over gloo) leaves the gradient image bit-identical to the single all-reduce of the whole image."""
    out = run_ranks(_bucket_rank, 2)
    assert out == {0: True, 1: True}, out


class _FakeSampler:
    """Stands in for the drop-in diffusion object in the host-logic test of sample_all_sharded (no GPU in this tier):
    same sample_all signature; in "philox" mode a row is a pure function of (seed, global row index), as the kernels'
    counters make it; in "torch" mode rows come from torch's global generator, as the reference's draws do."""

    def __init__(self, rng):
        self.rng = rng
        self.alphas = torch.zeros(1)
        self.calls = []

    def sample_all(self, num_samples, batch_size, target_dist, ddim=False, row0=0, seed=None, _targets=None):
        self.calls.append((num_samples, batch_size, row0, seed))
        if num_samples == 0:
            return torch.empty((0, 3), dtype=torch.float64), torch.empty((0,), dtype=torch.long)
        if _targets is not None:  # the caller's labels, one per row of this call (GaussianMultinomialDiffusion.sample_all does the same)
            assert _targets.shape == (num_samples,)
            rows = torch.arange(row0, row0 + num_samples, dtype=torch.float64)
            return torch.stack([rows, torch.full_like(rows, float(seed % 1000)), (rows * 7 + seed % 13) % 5], 1), _targets.long()
        if self.rng == "philox":
            rows = torch.arange(row0, row0 + num_samples, dtype=torch.float64)
            x = torch.stack([rows, torch.full_like(rows, float(seed % 1000)), (rows * 7 + seed % 13) % 5], 1)
        else:
            x = torch.randn((num_samples, 3), dtype=torch.float64)
        return x, torch.multinomial(target_dist, num_samples, replacement=True)


def _sample_rank(rank, ws, n, rng):
    import midst_toolkit_amd.distributed as D

    torch.manual_seed(1234)  # the reference pipeline seeds every process identically (common/random.py:11-55)
    fake = _FakeSampler(rng)
    x, y = D.sample_all_sharded(fake, n, 16, torch.ones(4))
    return x.numpy(), y.numpy(), fake.calls  # numpy: a tensor in the queue would be shared through a file descriptor of a process that has exited


@pytest.mark.parametrize("n", [37, 1])
def test_sample_all_sharded_philox_rows_do_not_depend_on_world_size(n):
    out = run_ranks(_sample_rank, 2, n, "philox")
    x0, y0, calls0 = (torch.from_numpy(out[0][0]), torch.from_numpy(out[0][1]), out[0][2])
    x1, y1, calls1 = (torch.from_numpy(out[1][0]), torch.from_numpy(out[1][1]), out[1][2])
    assert torch.equal(x0, x1) and torch.equal(y0, y1) and x0.shape == (n, 3) and y0.shape == (n,)
    assert calls0[0][3] == calls1[0][3], "every rank must use the seed broadcast from rank 0"
    assert calls0[0][2] == 0 and calls1[0][2] == (n + 1) // 2, "rank r's Philox rows start at its shard offset"
    assert torch.equal(x0[:, 0], torch.arange(n, dtype=torch.float64)), "gathered in global row order, each row once"
    # one process, same seed: the same table
    torch.manual_seed(1234)
    import midst_toolkit_amd.distributed as D

    single, y_single = D.sample_all_sharded(_FakeSampler("philox"), n, 16, torch.ones(4))
    assert torch.equal(single, x0)
    # the labels too: drawn once on rank 0 for the global row range, batch by batch as one process draws them - not one
    # identical sequence per identically seeded rank (which would gather into world_size copies of the same draws)
    torch.manual_seed(1234)
    torch.randint(1, 2**62, (1,), dtype=torch.int64)  # shared_seed()'s draw
    want = torch.cat([torch.multinomial(torch.ones(4), 16, replacement=True) for _ in range(-(-n // 16))])[:n]
    assert torch.equal(y0, want)
    if n > 8:
        h = (n + 1) // 2
        assert not torch.equal(y0[: n - h], y0[h:]), "the two shards' labels must not be copies of each other"


def test_sample_all_sharded_torch_rng_decorrelates_identically_seeded_ranks():
    out = run_ranks(_sample_rank, 2, 40, "torch")
    x0, x1 = torch.from_numpy(out[0][0]), torch.from_numpy(out[1][0])
    assert torch.equal(x0, x1) and x0.shape == (40, 3)
    # identically seeded processes used to produce the same rows on every rank: the two shards must differ
    assert not torch.equal(x0[:20], x0[20:])
    assert float((x0[:20] - x0[20:]).abs().min()) > 0
