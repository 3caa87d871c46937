#!/bin/bash
# usage: tools/pmc.sh <tag> <counters...>   (run on the GPU box through gpurun)
tag=$1; shift
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag=sys.argv[1]
for f in glob.glob(f"gpurun_out/pmc_{tag}/*/*counter_collection.csv"):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in acc.items():
        if "fused" in k or "gemm" in k or "rows" in k or "sample" in k:
            print(k, {a: f"{b:.4g}" for a,b in v.items()})
PY
