# timing-only ablations of the engine (results wrong): how much of a sweep's time does each phase of a step carry?
cd $GRAFT_REPO_ROOT
. tools/ab_scaffold.sh      # the experiment branches live in archive/proto/ablation_macros.patch since round 6
export MPG_BENCH_NO_F32=1
export MPG_FWD_CFLAGS="-mllvm -amdgpu-sched-strategy=max-memory-clause" MPG_BWD_CFLAGS="-mllvm -amdgpu-sched-strategy=max-memory-clause"
for V in "" "-DMPG_AB_NOMFMA" "-DMPG_AB_NO_IMGWRITE" "-DMPG_AB_NO_ELU8" "-DMPG_AB_NO_H1" "-DMPG_AB_NOMFMA -DMPG_AB_NO_IMGWRITE -DMPG_AB_NO_ELU8 -DMPG_AB_NO_H1" ""; do
  echo "== [$V]"; MPG_EXTRA_CFLAGS="$V" python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1 || { tail -3 /tmp/b.log | cut -c1-200; continue; }
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.4f' % d['ms_per_step'], 'bwd %.4f fwd %.4f' % (d['roofline']['avg_ms'], d['roofline_other_rollout_kernel']['avg_ms']), {k:round(v,4) for k,v in d['other_kernels_avg_ms'].items() if v and ('critic' in k or 'target' in k)})" 2>&1 | tail -1
done
python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1
