#!/bin/bash
# tools/kernel_resources.sh <file.hip> [grep-pattern] [extra hipcc flags...]: registers, scratch and spills per kernel
# as the compiler reports them (no GPU needed).
set -euo pipefail
src=$1; pat=${2:-.}; shift; shift || true
cd "$(dirname "$0")/../midst-toolkit_amd/csrc"
fp=""
case "$src" in mtk_diffusion.hip|mtk_tf.hip|mtk_train.hip|mtk_transform.hip|mtk_classifier.hip) fp="-ffp-contract=off";; esac
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. $fp "$@" -Rpass-analysis=kernel-resource-usage -c "$src" -o /tmp/kr_$$.o 2>&1 \
  | grep -E "remark: +(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize|SGPRs Spill|VGPRs Spill)" \
  | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
  | awk '/Function Name/{if(line)print line; sub(/Function Name: /,""); line=$0; next}{line=line" | "$0}END{print line}' | c++filt | grep -E "$pat" || true
rm -f /tmp/kr_$$.o
