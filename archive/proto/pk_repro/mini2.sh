#!/bin/bash
# builds and runs the variants of mini2.hip on the GPU box:   bash archive/proto/pk_repro/mini2.sh [launches] [neighbour iterations]
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}/archive/proto/pk_repro
for V in ${VARIANTS:-"" "-DPK=0" "-DNEIGHBOR=0" "-DK_EXEC=0" "-DK_LDS=0" "-DK_GLOBAL=0" "-DSTEPS=1" "-DK_LDS=0_-DK_EXEC=0_-DSTEPS=1" "-DK_LDS=0_-DK_EXEC=0_-DSTEPS=1_-DK_COPY=1" "-DK_LDS=0_-DK_EXEC=0_-DSTEPS=1_-DK_NOP=1" "-DK_EXEC=0_-DSTEPS=1_-DK_STORE_NOP=1" "-DK_EXEC=0_-DSTEPS=1_-DK_LDSWAIT=1" "-DK_GLOBAL=2" "-DK_GLOBAL=2_-DK_LDS=0_-DK_EXEC=0"}; do
  V=${V//_-D/ -D}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize $V -o /tmp/mini2 mini2.hip > /tmp/mini2.log 2>&1 || { echo "BUILD FAILED [$V]"; grep error /tmp/mini2.log | head -3; continue; }
  timeout 300 /tmp/mini2 ${1:-300} ${2:-600} ${3:-0}
done
