#!/bin/bash
# bash archive/proto/agpr4/run.sh   (on the GPU box; builds into gpurun_out/)
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Impg_amd/csrc -Wno-unused-result -mllvm -disable-promote-alloca-to-lds -fno-slp-vectorize \
    -save-temps=obj archive/proto/agpr4/fwd_a4.hip -o gpurun_out/fwd_a4 > gpurun_out/fwd_a4_build.log 2>&1 || { tail -20 gpurun_out/fwd_a4_build.log; exit 1; }
./gpurun_out/fwd_a4
