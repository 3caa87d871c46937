#!/bin/bash
# Builds libmtk_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
INC=../../include
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$INC -I. -Wall -Wno-unused-function"
mkdir -p _build
pids=()
# the diffusion kernels mirror eager fp32 op sequences: no fused multiply-add contraction
hipcc $FLAGS -ffp-contract=off -c mtk_diffusion.hip -o _build/mtk_diffusion.o & pids+=($!)
hipcc $FLAGS -ffp-contract=off -c mtk_tf.hip -o _build/mtk_tf.o & pids+=($!)
hipcc $FLAGS -ffp-contract=off -c mtk_train.hip -o _build/mtk_train.o & pids+=($!)
hipcc $FLAGS -c mtk_capi.hip -o _build/mtk_capi.o & pids+=($!)
hipcc $FLAGS -c mtk_mlp_f32.hip -o _build/mtk_mlp_f32.o & pids+=($!)
hipcc $FLAGS -c mtk_mlp_bf16.hip -o _build/mtk_mlp_bf16.o & pids+=($!)
hipcc $FLAGS -c mtk_mlp_bf16_layered.hip -o _build/mtk_mlp_bf16_layered.o & pids+=($!)
hipcc $FLAGS -ffp-contract=off -c mtk_transform.hip -o _build/mtk_transform.o & pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC _build/*.o -o ../libmtk_hip.so
echo "built $(cd .. && pwd)/libmtk_hip.so"
