#!/bin/bash
# bash archive/proto/pingpong/run.sh   (on the GPU box; builds into gpurun_out/)
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}
mkdir -p gpurun_out
for f in fwd_pp; do
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Impg_amd/csrc -Wno-unused-result -mllvm -disable-promote-alloca-to-lds \
    archive/proto/pingpong/$f.hip -o gpurun_out/$f > gpurun_out/${f}_build.log 2>&1 || { tail -20 gpurun_out/${f}_build.log; exit 1; }
./gpurun_out/$f
done
