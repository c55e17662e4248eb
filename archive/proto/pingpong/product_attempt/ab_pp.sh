cd $GRAFT_REPO_ROOT
echo "== ping-pong build"; python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1; python3 tools/pp_bench.py 2>/dev/null
echo "== lock-step build"; MPG_EXTRA_CFLAGS=-DMPG_PP_MIN_GROUPS_PER_WG=1000000 python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1; python3 tools/pp_bench.py 2>/dev/null
python3 -m mpg_amd.build --split-only > /tmp/b.log 2>&1
