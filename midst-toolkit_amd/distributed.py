"""Host-side sharding for the three entry points.  One process per GPU; the only data-path
collectives are the gather of results (scoring, sampling) and the gradient all-reduce (training, in trainer.py).

Works on any ``torch.distributed`` backend: ``nccl`` (= RCCL over xGMI) on the GPU box, ``gloo`` in the CPU tests.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist
from torch import Tensor


def world() -> tuple[int, int]:
    """(rank, world_size); (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of ``n`` units for ``rank``: the first ``n % world`` ranks hold one extra unit.
    ``lo`` is also the ``row0`` handed to the kernels, which makes in-kernel Philox noise independent of the
    number of GPUs."""
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_rows(local: Tensor, n_total: int, dim: int = 0, force_collective: bool = False) -> Tensor:
    """All-gather blocks that were split with :func:`shard_range` along ``dim`` back into global order.
    Equal blocks go through one ``all_gather_into_tensor``; ragged ones are padded to the largest block.
    ``force_collective``: go through the collective even at world size 1 (the one-GPU rehearsal of the RCCL path)."""
    rank, ws = world()
    if ws == 1 and not (force_collective and dist.is_available() and dist.is_initialized()):
        return local
    local = local.movedim(dim, 0).contiguous()
    sizes = [shard_range(n_total, r, ws)[1] - shard_range(n_total, r, ws)[0] for r in range(ws)]
    assert local.shape[0] == sizes[rank], (local.shape, sizes, rank)
    big = max(sizes)
    if big != local.shape[0]:
        pad = torch.zeros((big - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    out = torch.empty((ws * big,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local)
    if min(sizes) != big:
        out = torch.cat([out[r * big: r * big + sizes[r]] for r in range(ws)], 0)
    return out.movedim(0, dim)


@torch.no_grad()
def score_records_sharded(diffusion_model, features: Tensor, noise: Tensor, timesteps, additional_timestep: int = 0,
                          precision: str | None = None, force_collective: bool = False) -> Tensor:
    """:func:`scoring.score_records` over all ranks: every rank passes the SAME ``features`` ([N, .], host or device),
    scores its contiguous block on its GPU and receives the full [n_t, N, P] score tensor."""
    from .scoring import score_records

    rank, ws = world()
    lo, hi = shard_range(features.shape[0], rank, ws)
    dev = diffusion_model.alphas.device
    local = score_records(diffusion_model, features[lo:hi].to(dev), noise, timesteps, additional_timestep, precision)
    return gather_rows(local, features.shape[0], dim=1, force_collective=force_collective)


def _comm_device() -> torch.device:
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def shared_seed(force_collective: bool = False) -> int:
    """One 62-bit seed that every rank agrees on: drawn on rank 0 from torch's CPU generator and broadcast."""
    rank, ws = world()
    seed = torch.randint(1, 2**62, (1,), dtype=torch.int64)
    if ws > 1 or (force_collective and dist.is_available() and dist.is_initialized()):
        seed = seed.to(_comm_device())
        dist.broadcast(seed, src=0)
    return int(seed.item())


def shared_labels(num_samples: int, batch_size: int, target_dist: Tensor, force_collective: bool = False) -> Tensor:
    """The labels ``sample_all(num_samples, batch_size, target_dist)`` draws on one device (GMD:1401 per batch: one
    ``torch.multinomial`` of ``batch_size`` labels, the batches concatenated and cut to ``num_samples``), drawn on rank 0 from
    ITS generator and broadcast: label r belongs to global row r, so the gathered labels - like the rows - do not depend on the
    number of GPUs.  (Drawn per rank instead, identically seeded processes - the reference pipeline's habit - would all draw
    the same sequence, and the gathered vector would hold world_size copies of it.)"""
    rank, ws = world()
    n_batches = -(-num_samples // batch_size) if num_samples > 0 else 0
    if rank == 0:
        parts = [torch.multinomial(target_dist, num_samples=batch_size, replacement=True) for _ in range(n_batches)]
        y = (torch.cat(parts) if parts else torch.empty((0,), dtype=torch.long))[:num_samples].long()
    else:
        y = torch.empty((num_samples,), dtype=torch.long)
    if ws > 1 or (force_collective and dist.is_available() and dist.is_initialized()):
        y = y.to(_comm_device())
        dist.broadcast(y, src=0)
    return y.cpu()


def sample_all_sharded(diffusion_model, num_samples: int, batch_size: int, target_dist: Tensor, ddim: bool = False,
                       force_collective: bool = False):
    """``sample_all`` (GMD:1426) with the rows split over ranks; each rank runs its own T-loop, the only exchange is
    the final gather of the [N, Nn + C] samples and their labels.

    The reference pipeline seeds every process identically (common/random.py:11-55), so the ranks must be decorrelated
    here, not by the caller: with ``rng="philox"`` all ranks share ONE seed (broadcast from rank 0) and rank r's rows
    take the Philox row range [lo_r, hi_r), and the labels of all rows are drawn once on rank 0 (:func:`shared_labels`) - samples
    and labels do not depend on the number of GPUs; with ``rng="torch"`` each
    rank re-seeds torch's generators inside a fork with (a seed drawn from the caller's generator) + rank, and the
    caller's generator state is restored on return.  An empty shard contributes zero rows."""
    rank, ws = world()
    comm = ws > 1 or (force_collective and dist.is_available() and dist.is_initialized())
    lo, hi = shard_range(num_samples, rank, ws)
    if diffusion_model.rng == "philox":
        seed = shared_seed(force_collective)
        labels = shared_labels(num_samples, batch_size, target_dist, force_collective) if comm else None
        x, y = diffusion_model.sample_all(hi - lo, batch_size, target_dist, ddim=ddim, row0=lo, seed=seed,
                                          _targets=labels[lo:hi] if labels is not None else None)
    elif ws == 1:
        x, y = diffusion_model.sample_all(hi - lo, batch_size, target_dist, ddim=ddim)
    else:
        base = int(torch.randint(1, 2**31, (1,)).item())  # same on every rank when the processes were seeded alike
        dev = diffusion_model.alphas.device
        with torch.random.fork_rng(devices=[dev] if dev.type == "cuda" else []):
            torch.manual_seed(base + rank)
            x, y = diffusion_model.sample_all(hi - lo, batch_size, target_dist, ddim=ddim)
    if not comm:
        return x, y
    dev = diffusion_model.alphas.device if dist.get_backend() == "nccl" else torch.device("cpu")
    return (gather_rows(x.to(dev), num_samples, force_collective=force_collective).cpu(),
            gather_rows(y.to(dev), num_samples, force_collective=force_collective).cpu())


def allreduce_gradients(grad_image: Tensor) -> None:
    """Training's one exchange per iteration: SUM over ranks of the gradient image (each rank scaled its
    contribution by 1 / global batch in mtk_train_step)."""
    if world()[1] > 1:
        dist.all_reduce(grad_image)


class BucketedGradientExchange:
    """The gradient exchange of data-parallel training, one all-reduce per gradient bucket, overlapped with the backward
    (north_star part 1).  The library records an event per bucket as the backward finishes it
    (mtk_train_set_bucket_events); ``launch()`` - called right after the asynchronous backward call returns, while the GPU is
    still working through it - queues on a side stream, bucket by bucket in completion order, "wait for the bucket's event, then
    all-reduce (RCCL) that slice of the gradient image"; ``wait()`` makes the compute stream wait for the side stream before the
    optimizer kernel reads the image.  The sums are those of the single blocking all-reduce of the whole image: the same
    elements are added across the same ranks, only in several calls.

    On a CPU image or with the gloo backend (tests; two ranks sharing one GPU) the same bucket loop runs blocking."""

    def __init__(self, engine, grad_image: Tensor, layout: list[tuple[int, int]] | None = None, overlapped: bool | None = None):
        """``overlapped``: None = overlap whenever it applies (more than one rank, RCCL, a device image) unless the environment
        says ``MTK_DP_OVERLAP=0`` (the switch back to the blocking per-bucket all-reduce); True = also at world size 1 (the
        one-GPU rehearsal of the side-stream path); False = blocking."""
        self.image = grad_image
        self.layout = layout if layout is not None else (engine.train_bucket_layout() if engine is not None else [(0, grad_image.numel())])
        self.world = world()[1]
        can = bool(grad_image.is_cuda and dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl" and engine is not None)
        if overlapped is None:
            overlapped = self.world > 1 and os.environ.get("MTK_DP_OVERLAP", "1") != "0"
        elif overlapped and not can:
            raise RuntimeError("BucketedGradientExchange(overlapped=True) needs an initialised nccl (RCCL) process group and a device image")
        self.overlapped = bool(overlapped and can)
        self.forced = bool(self.overlapped and self.world == 1)
        self.engine = engine
        self.events = None
        self.stream = None
        if self.overlapped:
            self.events = [torch.cuda.Event() for _ in self.layout]
            engine.train_set_bucket_events(self.events)
            self.stream = torch.cuda.Stream(device=grad_image.device)

    def launch(self) -> None:
        if self.world == 1 and not self.forced:
            return
        if not self.overlapped:
            if self.image.is_cuda:
                torch.cuda.synchronize(self.image.device)
            for off, ln in self.layout:
                dist.all_reduce(self.image[off:off + ln])
            return
        with torch.cuda.stream(self.stream):
            for ev, (off, ln) in zip(self.events, self.layout):
                self.stream.wait_event(ev)
                dist.all_reduce(self.image[off:off + ln])

    def wait(self) -> None:
        if self.overlapped:
            torch.cuda.current_stream(self.image.device).wait_stream(self.stream)

    def close(self) -> None:
        if self.overlapped and self.engine is not None:
            torch.cuda.synchronize(self.image.device)
            self.engine.train_set_bucket_events(None)
            self.overlapped = False
