#!/bin/bash
cd $GRAFT_REPO_ROOT
MPG_EXTRA_CFLAGS="-DMPG_STAMP" python -m mpg_amd.build > /tmp/build.log 2>&1 || { tail -20 /tmp/build.log; exit 1; }
python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | grep -E "stamp|ms_per_step" | cut -c1-300 | tail -8
