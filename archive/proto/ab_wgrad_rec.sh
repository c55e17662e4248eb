#!/bin/bash
# Round 6: the dW2-only weight-gradient launch of the large batches with its H1 operand RECOMPUTED from the layer-1 inputs (shipped) against
# the control that reads it from the stash (-DMPG_WGRAD_FROM_STASH): parity of configs 3 / 4, then the C3 / C4 side lines of both builds on
# the same box.  Leaves the tree built with the shipped flags.   bash tools/ab_wgrad_rec.sh
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
export MPG_BENCH_NO_F32=1
P='import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith(chr(123))][-1]); print("%s ms/step %.4f regions %s" % (sys.argv[1], d["ms_per_step"], ["%.4f" % x for x in d["region_ms_per_step"]]))'
for v in "" "-DMPG_WGRAD_FROM_STASH" ""; do
  echo "== build [$v]"
  MPG_EXTRA_CFLAGS="$v" python3 -m mpg_amd.build > /dev/null 2>&1
  if [ -z "$v" ]; then timeout 900 python3 -m pytest tests/test_config34_gpu.py tests/test_noise_gpu.py -x -q -m gpu 2>&1 | tail -2; fi
  for c in c3 c4; do for i in 1 2; do python3 bench.py --config $c --no-cpu-baseline 2>/dev/null | python3 -c "$P" "$c [$v]"; done; done
done
