#!/bin/bash
# tools/profile_round.sh <tag>: on the GPU box (through gpurun), the rocprofv3 evidence of a round:
#   1. the default `python3 bench.py` command, plain (-> bench.json) and under --kernel-trace --stats (-> bench_*_kernel_stats.csv)
#   2. separate --pmc passes (FETCH_SIZE; WRITE_SIZE) of the same command, short    -> gpurun_out/<tag>/pmc_bench_{fetch,write}/
#   3. --kernel-trace --stats of benchmarks/bench_train_sample.py --train-only      -> gpurun_out/<tag>/train_*_kernel_stats.csv
#   4. --pmc passes of benchmarks/bench_kernels.py (the three HBM kernels)          -> gpurun_out/<tag>/pmc_kernels_{fetch,write}/
#   5. the training iteration at config 5's per-rank shape (2,048 x 4x1024, bf16), kernel stats and plain timings
#   6. the layered 4x1024 denoiser in chain mode (one timestep per launch): kernel stats and --pmc FETCH_SIZE / WRITE_SIZE passes
#   7. small-batch reverse chains (1,024 ... 16,384 rows, eager vs graph; caller-sized vs general kernels; kernel stats at 1,024 rows)
#      and the state-input vs matrix-input chain
# Each step runs under its own time limit; a step that times out or aborts ends the script.
set -uo pipefail
tag=${1:-prof}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
step() {  # name seconds command...
  local name=$1 secs=$2; shift 2
  echo "== $name"
  timeout -s KILL "$secs" "$@" > "$out/$name.log" 2>&1
  local rc=$?
  echo "   rc=$rc $(tail -n 1 "$out/$name.log" | cut -c1-200)"
  case $rc in 124|137|134|139) echo "!! $name ended with $rc: stopping"; exit $rc;; esac
}
# the line itself without the profiler (kernel tracing costs the launch-bound extras up to 50 %), then the same command traced
step bench_plain 400 python3 "$root/bench.py"
grep '^{' "$out/bench_plain.log" | tail -n 1 > "$out/bench.json"
step bench_stats 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/bench_stats" -o bench -- python3 "$root/bench.py"
grep '^{' "$out/bench_stats.log" | tail -n 1 > "$out/bench_traced.json"
step pmc_bench_fetch 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_bench_fetch" -o pmc -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras
step pmc_bench_write 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_bench_write" -o pmc -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras
step train_stats 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/train_stats" -o train -- python3 "$root/benchmarks/bench_train_sample.py" --train-only
step train1024_stats 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/train1024_stats" -o train1024 -- python3 "$root/benchmarks/bench_train_sample.py" --train-only --hidden 1024 --batch 2048 --train-precision bf16
step train1024 200 python3 "$root/benchmarks/bench_train_sample.py" --train-only --hidden 1024 --batch 2048 --train-precision bf16
step train1024_fp32 200 python3 "$root/benchmarks/bench_train_sample.py" --train-only --hidden 1024 --batch 2048
step layered_stats 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/layered_stats" -o layered -- python3 "$root/benchmarks/bench_layered.py" --rows 262144 --one-t
step pmc_layered_fetch 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_layered_fetch" -o pmc -- python3 "$root/benchmarks/bench_layered.py" --rows 262144 --one-t
step pmc_layered_write 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_layered_write" -o pmc -- python3 "$root/benchmarks/bench_layered.py" --rows 262144 --one-t
step graph_chain_512 300 python3 "$root/benchmarks/bench_graph.py" --hidden 512 --rows 1024,2048,4096,8192,16384
step graph_chain_1024 300 python3 "$root/benchmarks/bench_graph.py" --hidden 1024 --rows 1024,2048,4096,8192,16384
# the same chains with the general kernels at every batch size (A/B switches): the "before" of the caller-sized kernels, same box
step graph_chain_512_general 300 env MTK_LAYERED_BIG=1 MTK_STEP_ROWS=1 python3 "$root/benchmarks/bench_graph.py" --hidden 512 --rows 1024,8192
step graph_chain_1024_general 300 env MTK_LAYERED_BIG=1 MTK_STEP_ROWS=1 python3 "$root/benchmarks/bench_graph.py" --hidden 1024 --rows 1024,8192
step small_chain_stats 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/small_chain_stats" -o small_chain -- python3 "$root/benchmarks/bench_graph.py" --hidden 1024 --rows 1024
step chain_state 300 python3 "$root/benchmarks/bench_chain.py"
step kernels_stats 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kernels_stats" -o kernels -- python3 "$root/benchmarks/bench_kernels.py"
step pmc_kernels_fetch 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_kernels_fetch" -o pmc -- python3 "$root/benchmarks/bench_kernels.py" --hbm-only
step pmc_kernels_write 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_kernels_write" -o pmc -- python3 "$root/benchmarks/bench_kernels.py" --hbm-only
# summaries: per kernel, launches and mean counter value
python3 - "$out" <<'PY'
import collections, csv, glob, json, os, sys
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:90]
            if "at::" in k or "rocclr" in k or "elementwise" in k:
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    res[os.path.basename(d)] = {k: {"launches": len(n[k]), **{c: v / len(n[k]) for c, v in cs.items()}} for k, cs in acc.items()}
json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(res)[:3000])
PY
for f in $(find "$out" -name "*kernel_stats.csv"); do cp "$f" "$out/$(basename $(dirname $(dirname $f)))_$(basename $f)" 2>/dev/null || true; done
ls "$out"
