#!/bin/bash
# The FIRST multi-GPU run as one command (BASELINE.json configs[4]: MPG n = 25, 8 x 4096 rows, gradient exchanged
# between the GPUs).  Nothing here has ever run on more than one device: this script is the runbook, its output the first evidence.
#
#   bash tools/first_8gpu.sh            # on a node with 2 .. 8 MI355X; writes gpurun_out/first_8gpu/{*.log,*.json,table.txt}
#
# 1. tests/test_dist_gpu.py with ONE RANK PER DEVICE (MPG_TEST_RANK_PER_DEVICE=1: the IPC staging arrays, interprocess events and
#    copies then really cross xGMI; on the 1-GPU box the same tests time-share device 0);
# 2. bench.py --gpus N for N in 1 2 4 8 (as far as the node goes) x exchange form {nccl (RCCL), oneshot, twoshot} x
#    MPG_OVERLAP_EXCHANGE {0, 1} (the critics' slice exchanged under the reverse sweep), 200 steps x 5 regions each;
# 3. config 5's strong-scaling anchor: its GLOBAL batch (32 768 rows) on ONE GPU (bench.py --rows-per-gpu 32768);
# 4. one table: ms/step, whole-job env-steps/s, weak-scaling efficiency against N = 1, exchange_ms inside the step, the stand-alone
#    exchange per form, and config 5's strong-scaling speed-up = (32 768 rows on 1 GPU) / (8 x 4096 rows on 8 GPUs).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0 MPG_BENCH_NO_F32=1
OUT=gpurun_out/first_8gpu
mkdir -p "$OUT"
NDEV=$(python3 -c "import torch; print(torch.cuda.device_count())")
echo "devices: $NDEV" | tee "$OUT/table.txt"
python3 -m mpg_amd.build > "$OUT/build.log" 2>&1 || { echo "build failed, see $OUT/build.log"; exit 1; }

echo "== 1. tests/test_dist_gpu.py, one rank per device ==" | tee -a "$OUT/table.txt"
MPG_TEST_RANK_PER_DEVICE=1 timeout 3000 python3 -m pytest tests/test_dist_gpu.py -q -m gpu > "$OUT/dist_tests.log" 2>&1
tail -3 "$OUT/dist_tests.log" | tee -a "$OUT/table.txt"

echo "== 2. bench.py --gpus N x exchange form x overlap ==" | tee -a "$OUT/table.txt"
run() {   # name, gpus, then env assignments
    local name=$1 n=$2; shift 2
    env "$@" timeout 1500 python3 bench.py --gpus "$n" --steps 200 --warmup 20 --no-cpu-baseline --no-side-configs \
        > "$OUT/$name.json" 2> "$OUT/$name.err" || echo "$name: rc $?" | tee -a "$OUT/table.txt"
}
run n1 1 MPG_DIST_BACKEND=nccl
for n in 2 4 8; do
    [ "$n" -le "$NDEV" ] || continue
    for ov in 0 1; do
        run "n${n}_nccl_ov$ov"    "$n" MPG_DIST_BACKEND=nccl MPG_OVERLAP_EXCHANGE=$ov
        run "n${n}_oneshot_ov$ov" "$n" MPG_DIST_BACKEND=oneshot MPG_ONESHOT_MODE=oneshot MPG_OVERLAP_EXCHANGE=$ov
        run "n${n}_twoshot_ov$ov" "$n" MPG_DIST_BACKEND=oneshot MPG_ONESHOT_MODE=twoshot MPG_OVERLAP_EXCHANGE=$ov
    done
    # control of round 6's hand-shake: the separate "reads done" events of rounds 4 - 5 (EXPERIMENTS.md 6.2)
    run "n${n}_oneshot_ov0_Sevents" "$n" MPG_DIST_BACKEND=oneshot MPG_ONESHOT_MODE=oneshot MPG_OVERLAP_EXCHANGE=0 MPG_ONESHOT_S_EVENTS=1
done
echo "== 3. config 5's global batch on one GPU ==" | tee -a "$OUT/table.txt"
timeout 1500 python3 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-side-configs --rows-per-gpu 32768 \
    > "$OUT/c5_on_1gpu.json" 2> "$OUT/c5_on_1gpu.err"

python3 - "$OUT" <<'PY' | tee -a "$OUT/table.txt"
import glob, json, os, sys
out = sys.argv[1]
def load(name):
    try:
        lines = [l for l in open(os.path.join(out, name + '.json')).read().splitlines() if l.startswith('{')]
        return json.loads(lines[-1])
    except Exception:
        return None
base = load('n1')
print('%-22s %4s %10s %14s %9s %12s  %s' % ('run', 'N', 'ms/step', 'env-steps/s', 'weak eff', 'exchange_ms', 'stand-alone exchange forms (ms)'))
for f in sorted(glob.glob(os.path.join(out, 'n*.json')), key=lambda p: (len(os.path.basename(p)), p)):
    name = os.path.basename(f)[:-5]
    d = load(name)
    if not d:
        print('%-22s  no JSON line (see %s.err)' % (name, name))
        continue
    eff = d['value'] / (d['n_gpus'] * base['value']) if base else float('nan')
    print('%-22s %4d %10.4f %14.0f %9.3f %12s  %s' % (name, d['n_gpus'], d['ms_per_step'], d['value'], eff,
          '%.4f' % d['exchange_ms'] if d.get('exchange_ms') else '-', json.dumps(d.get('exchange_forms_ms'))))
c5 = load('c5_on_1gpu')
if c5:
    print('config 5 global batch (32 768 rows) on ONE GPU: %.4f ms/step, %.0f env-steps/s; sweeps %.3f + %.3f ms at %.2f / %.2f of the HBM roof'
          % (c5['ms_per_step'], c5['value'], c5['roofline']['avg_ms'], c5['roofline_other_rollout_kernel']['avg_ms'],
             c5['roofline']['frac_hbm'], c5['roofline_other_rollout_kernel']['frac_hbm']))
    for form in ('nccl', 'oneshot', 'twoshot'):
        d8 = load('n8_%s_ov0' % form)
        if d8:
            print('config 5 strong scaling, %s: 8 GPUs x 4096 rows %.4f ms/step -> speed-up %.2f of 8 over one GPU at 32 768 rows'
                  % (form, d8['ms_per_step'], c5['ms_per_step'] / d8['ms_per_step']))
PY
echo "table: $OUT/table.txt"
