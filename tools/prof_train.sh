#!/bin/bash
# tools/prof_train.sh <tag> [bench args...]: rocprofv3 kernel stats of the training-iteration bench into gpurun_out/<tag>/
set -euo pipefail
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o prof -- python3 "$root/benchmarks/bench_train_sample.py" "$@" > "$out/bench.log" 2>&1
f=$(find "$out" -name '*kernel_stats.csv' | head -1)
head -25 "$f" | cut -c1-220
