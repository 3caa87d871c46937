"""Error of the guidance classifier's split-bf16 mode against its fp32 mode (random-init ClavaDDPM default classifier)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import midst_toolkit_amd as mtk
dev = torch.device("cuda:0")
torch.manual_seed(0)
Nn, n_cls, B = 20, 12, 8192
clf = mtk.Classifier(Nn, n_cls, 128, [128, 256, 512, 1024, 512, 256, 128], dropout_prob=0.1).to(dev).eval()
x = torch.randn((B, Nn), device=dev); t = torch.randint(0, 1000, (B,), device=dev); y = torch.randint(0, n_cls, (B,), device=dev)
with torch.no_grad():
    l0, g0 = clf(x, t), clf.guidance_gradient(x, t, y, 1.0)
    clf.precision = "bf16x3"
    l1, g1 = clf(x, t), clf.guidance_gradient(x, t, y, 1.0)
rel = ((g1 - g0).norm(dim=1) / g0.norm(dim=1).clamp_min(1e-30)).cpu().numpy()
print("grad rel L2 per row: median %.3g  p90 %.3g  p99 %.3g  max %.3g" % (np.median(rel), np.quantile(rel, .9), np.quantile(rel, .99), rel.max()))
print("grad max abs err / max abs: %.3g" % float((g1 - g0).abs().max() / g0.abs().max()))
print("cosine: min %.6f" % float(torch.nn.functional.cosine_similarity(g0, g1).min()))
print("logits max abs err / scale: %.3g" % float((l1 - l0).abs().max() / l0.abs().max()))
