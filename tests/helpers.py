"""Shared helpers for the oracle and GPU parity tests."""
from __future__ import annotations

import os

import numpy as np
import torch

from oracle import diffusion_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["A_numeric", "B_ragged", "M_mixed", "N_tf", "L_catonly", "K1_single"]
# fixtures generated from the reference constructed with GaussianParametrization.X0 / Parametrization.DIRECT (make_golden.py --param-only):
# tag -> the oracle's flags
PARAM_FLAGS = {"P_x0direct": {"x0_param": True, "direct": True}}


def oracle_flags(c: dict) -> dict:
    return dict(PARAM_FLAGS.get(c.get("tag", ""), {}))


def load_case(tag: str) -> dict:
    with np.load(os.path.join(GOLDEN, f"case_{tag}.npz")) as z:
        d = {k: z[k] for k in z.files}
    d["K"] = d["K"].astype(np.int64)
    d["Nn"] = int(d["Nn"])
    d["T"] = int(d["T"])
    d["sched"] = str(d["sched"])
    d["d_layers"] = [int(v) for v in d["d_layers"]]
    d["C"] = len(d["K"]) if int(d["K"][0]) != 0 else 0
    d["S"] = int(d["K"].sum())
    d["D"] = d["Nn"] + d["S"]
    d["tag"] = tag
    return d


def case_params(c: dict) -> dict[str, torch.Tensor]:
    return orc.init_mlp_params(c["D"], c["d_layers"], seed=0)


def tt(a, dtype=None) -> torch.Tensor:
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(dtype) if dtype is not None else t


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double(), b.double()
    return float(((a - b).abs() / b.abs().clamp_min(1e-6)).max()) if a.numel() else 0.0


def load_guided(tag: str = "A_numeric") -> dict:
    """Fixture of tools/make_golden.py::gen_guided: the reference's conditional_sample ("cs") and guided sample ("sg")."""
    with np.load(os.path.join(GOLDEN, f"guided_{tag}.npz")) as z:
        d = {k: z[k] for k in z.files}
    d["Nn"], d["T"], d["scale"] = int(d["Nn"]), int(d["T"]), float(d["scale"])
    d["d_layers"] = [int(v) for v in d["d_layers"]]
    d["params"] = {k[2:]: tt(v) for k, v in d.items() if k.startswith("w_")}
    return d


def guidance_function(centers: torch.Tensor, scale: float, T: int):
    """The closed-form stand-in for the classifier gradient that generated the guided fixture (same formula as
    tools/make_golden.py::guidance_function); device-agnostic."""
    def conditioning_function(features, timestep, **kwargs):
        y = kwargs["y"].view(-1).long().to(features.device)
        x = features.detach().float()
        return (centers.to(features.device)[y] - x) * (scale * (1.0 + timestep.float()[:, None] / T))
    return conditioning_function


def load_ddim(tag: str) -> dict:
    """Fixture of tools/make_golden.py::gen_ddim: the reference's sample_ddim with every draw captured."""
    with np.load(os.path.join(GOLDEN, f"ddim_{tag}.npz")) as z:
        d = {k: z[k] for k in z.files}
    d["K"] = d["K"].astype(np.int64)
    d["Nn"], d["T"], d["sched"] = int(d["Nn"]), int(d["T"]), str(d["sched"])
    d["d_layers"] = [int(v) for v in d["d_layers"]]
    d["C"] = len(d["K"]) if int(d["K"][0]) != 0 else 0
    d["S"] = int(d["K"].sum())
    d["D"] = d["Nn"] + d["S"]
    d["tag"] = tag
    return d


def load_ddimx(tag: str = "A_numeric") -> dict:
    """Fixture of tools/make_golden.py::gen_ddim_family: gaussian_ddim_sample (eta > 0, with and without condition_score),
    sample_ddim with a conditioning function, gaussian_ddim_reverse_sample and multinomial_ddim_step (eta > 0) of the reference."""
    with np.load(os.path.join(GOLDEN, f"ddimx_{tag}.npz")) as z:
        d = {k: z[k] for k in z.files}
    d["Nn"], d["T"], d["md_T"] = int(d["Nn"]), int(d["T"]), int(d["md_T"])
    for k in ("scale", "eta", "eta_cs", "md_eta"):
        d[k] = float(d[k])
    d["d_layers"] = [int(v) for v in d["d_layers"]]
    d["K"], d["C"], d["S"], d["D"], d["sched"] = np.asarray([0], dtype=np.int64), 0, 0, d["Nn"], "cosine"
    d["md_K"] = d["md_K"].astype(np.int64)
    return d


def load_edm(tag: str = "latent24") -> dict:
    """Fixture of tools/make_golden.py::gen_edm: TabSyn's MLPDiffusion / Preconditioner / EDMLoss / sampler run by the reference."""
    with np.load(os.path.join(GOLDEN, f"edm_{tag}.npz")) as z:
        d = {k: z[k] for k in z.files}
    for k in ("d_in", "dim_t", "num_steps"):
        d[k] = int(d[k])
    d["params"] = {k[2:]: tt(v) for k, v in d.items() if k.startswith("w_")}
    return d


def load_label(tag: str = "label3") -> dict:
    """Fixture of tools/make_golden.py::gen_label_embedding: the reference's label-conditioned denoiser (IsTargetConditioned.EMBEDDING)."""
    with np.load(os.path.join(GOLDEN, f"label_{tag}.npz")) as z:
        d = {k: z[k] for k in z.files}
    for k in ("Nn", "T", "n_labels", "ch_T"):
        d[k] = int(d[k])
    d["K"] = np.asarray(d["K"], dtype=np.int64)
    d["d_layers"] = [int(v) for v in d["d_layers"]]
    d["params"] = {k[2:]: tt(v) for k, v in d.items() if k.startswith("p_")}
    return d


def load_classifier(tag: str = "A_numeric") -> dict:
    """Fixture of tools/make_golden.py::gen_classifier: the reference's own Classifier (eval mode) and conditioning function."""
    with np.load(os.path.join(GOLDEN, f"classifier_{tag}.npz")) as z:
        d = {k: z[k] for k in z.files}
    for k in ("Nn", "T", "n_classes"):
        d[k] = int(d[k])
    d["scale"] = float(d["scale"])
    d["d_layers"] = [int(v) for v in d["d_layers"]]
    d["hidden"] = [int(v) for v in d["hidden"]]
    d["params"] = {k[2:]: tt(v) for k, v in d.items() if k.startswith("w_")}
    d["state"] = {k[2:]: tt(v) for k, v in d.items() if k.startswith("c_")}
    return d
