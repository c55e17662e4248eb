#!/bin/bash
# One-GPU cost of the multi-GPU step beyond the bytes on the wire (round 5): the bench step (a) as shipped on one GPU, (b) through the
# exchange path in a ONE-rank RCCL group (the collective between mpg_step_begin and mpg_step_end + clip partials from the exchanged
# buffer), (c) the same with the critics' gradient finished ahead of the reverse sweep and exchanged on a second stream
# (MPG_OVERLAP_EXCHANGE=1).  bash tools/ab_exchange.sh
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
export MPG_BENCH_NO_F32=1
P='import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][-1]); o=d["other_kernels_avg_ms"]; print("ms/step %.4f  regions %s  exchange_ms %s  wgrad %.4f adam %.4f" % (d["ms_per_step"], ["%.4f" % x for x in d["region_ms_per_step"]], d["exchange_ms"], o["k_wgrad_multi"], o["k_clip_adam_polyak"]))'
for i in 1 2; do
  echo -n "(a) one GPU, no exchange:                "; python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-side-configs 2>/dev/null | python3 -c "$P"
  echo -n "(b) one-rank RCCL exchange:              "; python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-side-configs --always-exchange 2>/dev/null | python3 -c "$P"
  echo -n "(c) ... + critics under the sweep:       "; MPG_OVERLAP_EXCHANGE=1 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-side-configs --always-exchange 2>/dev/null | python3 -c "$P"
  # round 6: the one-shot IPC exchange with itself - the round-5 form (copy into the staging slot, sum, mpg_sq_partials) against the
  # slot form (the step writes its gradient straight into the slot; the sum leaves the clip partials: ONE launch more than (a))
  echo -n "(d) one-rank one-shot, round-5 form:     "; MPG_DIST_BACKEND=oneshot MPG_SLOT_EXCHANGE=0 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-side-configs --always-exchange 2>/dev/null | python3 -c "$P"
  echo -n "(e) one-rank one-shot, slot form:        "; MPG_DIST_BACKEND=oneshot python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-side-configs --always-exchange 2>/dev/null | python3 -c "$P"
done
