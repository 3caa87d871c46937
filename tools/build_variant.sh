#!/bin/bash
# A/B builds of the layered bf16 GEMM: tools/build_variant.sh <name> [-DFLAG ...] -> benchmarks/libmtk_<name>.so
# (the other objects come from midst-toolkit_amd/csrc/_build; run csrc/build.sh first).  Use with MTK_LIB_PATH.
set -euo pipefail
cd "$(dirname "$0")/../midst-toolkit_amd/csrc"
name=$1; shift
mkdir -p /tmp/mtk_variant_$name
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wall -Wno-unused-function "$@" -c mtk_mlp_bf16_layered.hip -o /tmp/mtk_variant_$name/layered.o
objs=$(ls _build/*.o | grep -v mtk_mlp_bf16_layered.o)
hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/mtk_variant_$name/layered.o -o ../../benchmarks/libmtk_$name.so
echo "built benchmarks/libmtk_$name.so"
