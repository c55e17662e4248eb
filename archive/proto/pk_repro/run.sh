#!/bin/bash
# builds the variants of repro.hip on the GPU box and prints their failure rates:   bash archive/proto/pk_repro/run.sh [launches=1000]
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}/archive/proto/pk_repro
N=${1:-1000}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-unused-value -mllvm -disable-promote-alloca-to-lds -I ../../../include -I ../../../mpg_amd/csrc"
run() {   # name, extra flags
  hipcc $FLAGS $2 -o /tmp/repro_$1 repro.hip > /tmp/repro_$1.log 2>&1 || { echo "$1: BUILD FAILED"; tail -3 /tmp/repro_$1.log; return; }
  printf "%-34s " "$1 [$2]"; timeout 300 /tmp/repro_$1 $N
}
run shipped "-DV_BOUNDS=4"
run pk "-DV_BOUNDS=4 -DMPG_AB_PKFMA"
run pk_1job "-DV_BOUNDS=4 -DMPG_AB_PKFMA -DV_JOBS=1"
run pk_2jobs "-DV_BOUNDS=4 -DMPG_AB_PKFMA -DV_JOBS=2"
run pk_3critics "-DV_BOUNDS=4 -DMPG_AB_PKFMA -DV_ALL_A"
run pk_nomfma "-DV_BOUNDS=4 -DMPG_AB_PKFMA -DMPG_AB_WG_NOMFMA"
run pk_1job_12288 "-DV_BOUNDS=4 -DMPG_AB_PKFMA -DV_JOBS=1 -DV_ROWS=12288"
run pk_1job_1024 "-DV_BOUNDS=4 -DMPG_AB_PKFMA -DV_JOBS=1 -DV_ROWS=1024"
# which ingredient: the packed FMAs, or the scratch spills the packed form happens to cause at the 128-register cap?
run pk_nospill "-DV_BOUNDS=4 -DMPG_AB_PKFMA -DMPG_AB_WG_W2_FIRST"
run shipped_spilling "-DV_BOUNDS=5"
run shipped_spilling_1job "-DV_BOUNDS=5 -DV_JOBS=1"
