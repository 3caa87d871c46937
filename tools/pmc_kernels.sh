#!/bin/bash
# usage: tools/pmc_kernels.sh <tag> <script.py and args, quoted> <counters...>   (on the GPU box through gpurun)
tag=$1; shift
script=$1; shift
export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -- python3 $script > gpurun_out/pmc_$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag=sys.argv[1]
for f in glob.glob(f"gpurun_out/pmc_{tag}/*/*counter_collection.csv"):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:70]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k,v in acc.items():
        if "at::" in k or "rocclr" in k: continue
        print(k, len(n[k]), {a: f"{b/len(n[k]):.4g}" for a,b in v.items()})
PY
