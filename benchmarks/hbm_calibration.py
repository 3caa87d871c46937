"""What the box's HBM gives plain streaming kernels (torch's own copy / fill / reduction), to read the diffusion kernels'
GB/s against: python3 benchmarks/hbm_calibration.py"""
import json
import torch

dev = torch.device("cuda:0")
n = 1 << 28  # 1 GiB of fp32 per tensor
x = torch.randn(n, device=dev)
y = torch.empty_like(x)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, fn, nbytes in (
    ("copy (read + write)", lambda: y.copy_(x), 8 * n),
    ("fill (write only)", lambda: y.fill_(1.0), 4 * n),
    ("sum (read only)", lambda: x.sum(), 4 * n),
    ("add_ scalar (read + write in place)", lambda: y.add_(1.0), 8 * n),
):
    ms = timed(fn)
    print(json.dumps({"op": name, "bytes": nbytes, "ms": ms, "GB/s": nbytes / ms / 1e6, "frac_of_8TBs": nbytes / ms / 1e6 / 8000}))
