#!/bin/bash
# Round 6: the one-shot / two-shot exchange with and without the separate "reads done" events (MPG_ONESHOT_S_EVENTS=1: rounds 4 - 5), two
# ranks time-sharing ONE GPU (on a multi-GPU node: set MPG_AB_RANKS and the ranks land on their own devices).  bash tools/ab_s_events.sh
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
export HSA_ENABLE_IPC_MODE_LEGACY=0 MPG_BENCH_NO_F32=1 MPG_BENCH_BURN_IN=60
N=${MPG_AB_RANKS:-2}
P='import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][-1]); print("%s ms/step %.4f exchange_ms %.4f regions %s" % (sys.argv[1], d["ms_per_step"], d["exchange_ms"] or 0, ["%.4f" % x for x in d["region_ms_per_step"]]))'
for rep in 1 2; do
for mode in oneshot twoshot; do
for s in 0 1; do
  MPG_DIST_BACKEND=oneshot MPG_ONESHOT_MODE=$mode MPG_ONESHOT_S_EVENTS=$s python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
    --master-port $((29600 + RANDOM % 300)) bench.py --gpus $N --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "$P" "$mode S_events=$s"
done; done; done
