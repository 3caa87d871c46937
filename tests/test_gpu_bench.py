"""bench.py's contract on the GPU: the one JSON line at N = 1, and a two-rank rehearsal of the N > 1 control flow
(double-buffered score blocks, asynchronous gather drained inside the timed region).  Two ranks cannot share one
card under RCCL, so the rehearsal runs the same loop with the gather staged through gloo (MTK_BENCH_BACKEND=gloo);
MTK_BENCH_CHECK makes every rank verify that its own block and every peer's arrived in rank order."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}


def _last_json(text):
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def test_bench_line_single_gpu():
    r = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--records", "8192", "--nt", "70", "--no-cpu-baseline",
                        "--sample-rows", "2048", "--sample-rows-large", "4096"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _last_json(r.stdout)
    assert KEYS <= set(out) and out["n_gpus"] == 1 and out["steps"] == 2 and out["value"] > 0
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert out["roofline"]["launches_per_step"] == 2  # 70 timesteps = one launch of 64 + one of 6
    assert "workload" in out["config"] and "math" in out["config"]["workload"]
    # the other half of the metric and the other modes ride in the same line, each with its own roofline fraction
    assert out["sampled_rows_per_s"] > 0 and out["sampled_rows_per_s_4x512"] > 0 and out["sampled_rows_per_s_1M_rows"] > 0
    ex = out["extra"]
    for k in ("sampling_4x512", "sampling_4x1024"):
        assert ex[k]["T"] == 1000 and 0 < ex[k]["roofline_denoiser"]["frac"] < 1 and 0 < ex[k]["roofline_step"]["frac"] < 1
    ml = ex["mixed_loss_shapeM"]
    assert ml["evals_per_s"] > 0 and all(0 < ml[k]["frac"] < 1 for k in ("roofline_q_sample", "roofline_loss_rows", "roofline_denoiser"))
    assert ex["parity_fp32"]["evals_per_s"] > 0 and ex["parity_bf16x3"]["evals_per_s"] > ex["parity_fp32"]["evals_per_s"]
    trn = ex["training_2048x4x1024"]  # one trainer iteration at configs[4]'s per-rank shape, both GEMM precisions
    assert 0 < trn["bf16"]["ms_per_step"] < trn["fp32"]["ms_per_step"] and trn["rows"] == 2048
    assert all(ex["sampling_guided_caller_sized"][k]["us_per_step"] > 0 for k in ("1024_rows", "8192_rows"))
    small = ex["sampling_caller_sized"]  # microseconds per reverse step at the batch sizes the reference's callers sample
    assert all(small[k]["us_per_step"] > 0 for k in ("4x512_1024_rows", "4x512_8192_rows", "4x1024_1024_rows", "4x1024_8192_rows",
                                                    "clavaddpm_default_512_1024x4_512_1024_rows", "clavaddpm_default_512_1024x4_512_8192_rows"))


def test_bench_two_rank_rehearsal():
    env = dict(os.environ, MTK_BENCH_BACKEND="gloo", MTK_BENCH_CHECK="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "2", "--records", "8192", "--nt", "2", "--sample-rows", "1024", "--sample-rows-large", "0"],
                       cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json(r.stdout)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["sampled_rows_per_s"] > 0  # every rank ran its own chain


def test_bench_two_rank_strong_scaling_rehearsal():
    """BASELINE configs[2] shape of the loop: a fixed record set split with shard_range (ragged: 8191 records over two ranks),
    P = 4, blocks gathered to rank 0."""
    env = dict(os.environ, MTK_BENCH_BACKEND="gloo", MTK_BENCH_CHECK="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29534", "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1", "--mode", "strong", "--records", "8191",
                        "--nt", "3"], cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json(r.stdout)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["noise_draws"] == 4
    assert out["config"]["evals_per_step"] == 8191 * 3 * 4


def test_bench_rccl_single_rank_rehearsal():
    """The N > 1 loop of bench.py on REAL RCCL, as far as one GPU allows: an nccl process group of one rank, device_id bound,
    the asynchronous double-buffered dist.gather of every step's score block, the barriers and the MAX all-reduce of the time
    (gloo cannot show that the RCCL calls and their stream semantics work; two ranks cannot share a card under RCCL)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, "bench.py", "--rehearse-rccl", "--steps", "5", "--warmup", "2", "--records", "8192", "--nt", "70",
                        "--no-cpu-baseline", "--sample-rows", "1024", "--sample-rows-large", "0"], cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _last_json(r.stdout)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["sampled_rows_per_s"] > 0
    assert "rehearse_rccl" in r.stderr
