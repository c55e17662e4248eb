#!/bin/bash
# Backend-knob lottery over the two sweeps' translation units: every variant = the shipped per-file flags
# (-mllvm -amdgpu-sched-strategy=iterative-ilp) + ONE more -mllvm option on rollout_fwd.hip and rollout_bwd.hip; same box, baseline
# first and last.      bash tools/ab_sched_lottery.sh            (through gpurun; KNOBS="..." overrides the list)
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
BASE="-mllvm -amdgpu-sched-strategy=iterative-ilp"
KNOBS=${KNOBS:-"-amdgpu-use-amdgpu-trackers -amdgpu-disable-unclustered-high-rp-reschedule -amdgpu-disable-clustered-low-occupancy-reschedule -amdgpu-set-wave-priority -amdgpu-early-ifcvt -amdgpu-opt-vgpr-liverange=0 -amdgpu-dce-in-ra=0 -amdgpu-enable-rewrite-partial-reg-uses=0 -amdgpu-load-store-vectorizer=0 -amdgpu-disable-loop-alignment -amdgpu-enable-pre-ra-optimizations=0 -enable-post-misched=0"}
ARGS=()
for k in $KNOBS; do ARGS+=("FWD=$BASE -mllvm $k BWD=$BASE -mllvm $k"); done
STEPS=${STEPS:-300} bash tools/ab.sh "${ARGS[@]}"
