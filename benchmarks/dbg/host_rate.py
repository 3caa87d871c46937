"""Host enqueue time vs device time of a training iteration (is the loop launch-bound?)."""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
import midst_toolkit_amd as mtk
dev = torch.device("cuda:0")
K, Nn = [10] * 10, 20
D = Nn + sum(K)
hidden, batch = int(sys.argv[1]), int(sys.argv[2])
net = mtk.MLPDiffusion(D, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([hidden] * 4, 0.0))
d = mtk.GaussianMultinomialDiffusion(np.asarray(K), Nn, net, num_timesteps=1000, device=dev).to(dev)
d.rng = "philox"
g = torch.Generator().manual_seed(1)
x0 = torch.cat([torch.randn((batch, Nn), generator=g)] + [torch.randint(0, k, (batch, 1), generator=g).float() for k in K], 1).to(dev)
def batches():
    while True:
        yield x0, None
tr = mtk.ClavaDDPMTrainer(d, batches(), 2e-4, 1e-4, 100, device=dev)
tr._engine.train_set_precision(sys.argv[3])
for _ in range(10):
    tr._train_step(x0)
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    tr._train_step(x0)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(json.dumps({"host_enqueue_ms_per_step": (t1 - t0) / n * 1e3, "total_ms_per_step": (t2 - t0) / n * 1e3}))
