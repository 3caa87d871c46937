"""Experiment: do two layered 4x1024 passes on two streams hide each other's launch ramp / drain?  Two engines (own workspaces), half the
rows each, on two torch streams, against one engine on one stream."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import midst_toolkit_amd as mtk

dev = torch.device("cuda:0")
D, H, B = 120, 1024, 262144
torch.manual_seed(0)
nets = [mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([H] * 4, 0.0)).to(dev) for _ in range(2)]
nets[1].load_state_dict(nets[0].state_dict())
for n in nets:
    n.precision = "bf16"
engs = [n.engine(dev) for n in nets]
x = torch.randn((D, B), device=dev)
out = torch.empty((D, B), device=dev)
h = B // 2
xa, xb = x[:, :h].contiguous(), x[:, h:].contiguous()
oa, ob = torch.empty((D, h), device=dev), torch.empty((D, h), device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def one():
    engs[0].mlp_forward(x, 500, out=out, precision="bf16", x_strides=(1, B), out_strides=(1, B), B=B)

def two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        engs[0].mlp_forward(xa, 500, out=oa, precision="bf16", x_strides=(1, h), out_strides=(1, h), B=h)
    with torch.cuda.stream(s2):
        engs[1].mlp_forward(xb, 500, out=ob, precision="bf16", x_strides=(1, h), out_strides=(1, h), B=h)
    cur.wait_stream(s1); cur.wait_stream(s2)

def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

for rep in range(2):
    print(json.dumps({"one_stream_ms": timeit(one), "two_streams_ms": timeit(two)}), flush=True)
two(); one(); torch.cuda.synchronize()
print("equal:", bool(torch.equal(out[:, :h], oa) and torch.equal(out[:, h:], ob)))
