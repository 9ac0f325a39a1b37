#!/bin/bash
# usage: tools/kernel_resources.sh csrc-file.hip [name-filter]
# compiles one HIP source for gfx950 and prints SGPRs / VGPRs / scratch bytes / spilled VGPRs per kernel
src="$1"; filt="${2:-.}"
cd "$(dirname "$0")/../carca_replication_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -c "$src" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name:/ {name=$5} /TotalSGPRs:/ {s=$4} / VGPRs:/ {v=$4} /ScratchSize/ {sc=$5} /SGPRs Spill:/ {ss=$5} /VGPRs Spill:/ {print name, "sgpr="s, "vgpr="v, "scratch="sc, "sspill="ss, "vspill="$5}' |
  c++filt | grep -E "$filt"
