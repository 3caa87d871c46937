#!/bin/bash
# runs bench_layered.py over a list of "lib:chunk" pairs (lib = name under benchmarks/libmtk_<name>.so, or "cur")
set -e
cd "$(dirname "$0")/.."
for pair in "$@"; do
  lib=${pair%%:*}; chunk=${pair##*:}
  if [ "$lib" = cur ]; then unset MTK_LIB_PATH; else export MTK_LIB_PATH=$PWD/benchmarks/libmtk_$lib.so; fi
  MTK_LAYERED_CHUNK=$chunk timeout -k 10 120 python benchmarks/bench_layered.py --tag "$pair"
done
