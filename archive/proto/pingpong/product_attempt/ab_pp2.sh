cd $GRAFT_REPO_ROOT
for V in "" "-DMPG_PP_AB_PLAINOUT" "-DMPG_PP_AB_NOCHECKS" "-DMPG_PP_AB_NOSTASH" "-DMPG_PP_AB_SIMPLEX" "-DMPG_PP_AB_PLAINOUT -DMPG_PP_AB_NOCHECKS -DMPG_PP_AB_NOSTASH -DMPG_PP_AB_SIMPLEX"; do
  echo "== [$V]"; MPG_EXTRA_CFLAGS="$V" python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1 || tail -3 /tmp/b.log; python3 tools/pp_bench.py 2>/dev/null | grep "in=8" | grep -E "65536|131072"
done
python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1
