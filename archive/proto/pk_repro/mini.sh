#!/bin/bash
# builds and runs the variants of mini.hip on the GPU box:   bash archive/proto/pk_repro/mini.sh
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}/archive/proto/pk_repro
for V in "-DMFMA_KIND=0" "-DMFMA_KIND=0 -DSEPARATE" "-DMFMA_KIND=1" "-DMFMA_KIND=2" "-DMFMA_KIND=3"; do
  hipcc --offload-arch=gfx950 -O3 $V -o /tmp/mini mini.hip > /tmp/mini.log 2>&1 || { echo "BUILD FAILED [$V]"; tail -3 /tmp/mini.log; continue; }
  timeout 300 /tmp/mini ${1:-100} ${2:-2000} ${3:-4000}
done
