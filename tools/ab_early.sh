MPG_WGRAD_EARLY=1 timeout 300 python -m pytest tests/test_learner_gpu.py -x -q -m gpu -k "golden or native_step or bench_size" 2>&1 | tail -2
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4), {k: round(v["ms_per_step"]*1e3,1) for k,v in d.get("kernel_groups_ms_per_step",{}).items()})'
for i in 1 2; do
MPG_WGRAD_EARLY=1 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "$P" early
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "$P" base
done
