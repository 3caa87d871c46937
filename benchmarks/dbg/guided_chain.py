"""Classifier-guided reverse chain (GMD:1319 conditional_sample; what ClavaDDPM runs for child tables) at caller-sized batches:
microseconds per step with and without the guidance classifier (ClavaDDPM's default classifier layers)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import midst_toolkit_amd as mtk

dev = torch.device("cuda:0")
Nn, T, n_cls = 20, 200, 12
torch.manual_seed(0)
net = mtk.MLPDiffusion(Nn, 0, mtk.IsTargetConditioned.NONE, mtk.DiffusionParameters([512, 1024, 1024, 1024, 1024, 512], 0.0))
net.precision = "bf16"
d = mtk.GaussianMultinomialDiffusion(np.asarray([0]), Nn, net, num_timesteps=T, device=dev).to(dev)
d.precision, d.rng, d.results_on_device = "bf16", "philox", True
clf = mtk.Classifier(Nn, n_cls, 128, [128, 256, 512, 1024, 512, 256, 128], dropout_prob=0.1).to(dev).eval()
clf.precision = os.environ.get("CLF_PRECISION", "fp32")
cond = mtk.get_conditioning_function(clf, 1.0)
for B in ([int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else (1024, 8192, 20000)):
    y = torch.randint(0, n_cls, (B,), device=dev)
    res = {}
    for guided in (False, True):
        def run():
            if guided:
                return d.conditional_sample(y.cpu(), model_kwargs={"y": y}, conditioning_function=cond)
            return d.sample(B, torch.ones(n_cls))
        run(); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(); torch.cuda.synchronize()
        res["guided" if guided else "plain"] = (time.perf_counter() - t0) / T * 1e6
    print(json.dumps({"classifier": clf.precision, "rows": B, "T": T, "us_per_step_plain": res["plain"], "us_per_step_guided": res["guided"]}), flush=True)
