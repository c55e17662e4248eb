cd $GRAFT_REPO_ROOT
. tools/ab_scaffold.sh      # the experiment branches live in archive/proto/ablation_macros.patch since round 6
mkdir -p gpurun_out/r04_tr
export MPG_BENCH_NO_F32=1
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d["other_kernels_avg_ms"]; print("ms/step %.4f (median %.4f) fwd %.4f bwd %.4f target %.4f critic %.4f wgrad %.4f pol %.4f env %.4f adam %.4f" % (d["ms_per_step"], d["step_ms_median"], d["roofline_other_rollout_kernel"]["avg_ms"], d["roofline"]["avg_ms"], o["k_target_fused"], o["k_critic_fused"], o["k_wgrad_multi"], o["k_forward (worker policy)"], o["k_step_store_reset (env)"], o["k_clip_adam_polyak"]))'
run() { python3 bench.py --steps 400 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "$P"; }
echo "== baseline (shipped flags)"; python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1; run
echo "== control: fwd with max-memory-clause, no TR"; MPG_FWD_CFLAGS="-mllvm -amdgpu-sched-strategy=max-memory-clause" python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1; run
echo "== TR image, fwd max-memory-clause"; MPG_EXTRA_CFLAGS=-DMPG_TR_IMAGE MPG_FWD_CFLAGS="-mllvm -amdgpu-sched-strategy=max-memory-clause" python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1 || tail -3 /tmp/b.log; run
python3 -m pytest tests/test_networks_gpu.py tests/test_rollout_gpu.py -m gpu -x -q 2>&1 | tail -3
python3 -m pytest tests/test_learner_gpu.py -m gpu -x -q -k "bench_size and split or repeated" 2>&1 | tail -3
echo "== TR image again"; run
echo "== baseline again"; python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1; run
