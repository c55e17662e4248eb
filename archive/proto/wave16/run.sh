#!/bin/bash
# bash archive/proto/wave16/run.sh   (on the GPU box; builds into gpurun_out/)
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Impg_amd/csrc -Wno-unused-result -mllvm -disable-promote-alloca-to-lds \
    archive/proto/wave16/fwd16.hip -o gpurun_out/fwd16 > gpurun_out/fwd16_build.log 2>&1 || { tail -20 gpurun_out/fwd16_build.log; exit 1; }
./gpurun_out/fwd16
