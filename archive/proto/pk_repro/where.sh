cd $GRAFT_REPO_ROOT/archive/proto/pk_repro
N=${1:-300}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-unused-value -mllvm -disable-promote-alloca-to-lds -I ../../../include -I ../../../mpg_amd/csrc"
run() {
  hipcc $FLAGS $2 -o /tmp/repro_$1 repro.hip > /tmp/repro_$1.log 2>&1 || { echo "$1: BUILD FAILED"; tail -3 /tmp/repro_$1.log; return; }
  printf "%-34s " "$1"; timeout 300 /tmp/repro_$1 $N | tail -1
}
B="-DV_BOUNDS=4 -DMPG_AB_PKFMA -DV_ALL_A -DV_JOBS=2 -DV_ROLES=1"
run neighbour_matrix_loop "$B"
run neighbour_mfma_regs "$B -DV_NEIGHBOR=1"
run neighbour_lds_barriers "$B -DV_NEIGHBOR=2"
run neighbour_global_loads "$B -DV_NEIGHBOR=3"
run neighbour_valu "$B -DV_NEIGHBOR=4"
run neighbour_mfma_regs_short "$B -DV_NEIGHBOR=1 -DV_NEIGHBOR_ITERS=300"
