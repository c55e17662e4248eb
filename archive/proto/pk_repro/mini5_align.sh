#!/bin/bash
# mini5.hip with the neighbour's MFMA loop head at 24, 28 and 0 mod 32 bytes
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}/archive/proto/pk_repro
LLVM=/opt/rocm/lib/llvm/bin
hipcc --offload-arch=gfx950 -O3 -o /tmp/mini5_host mini5.hip > /tmp/mini5_host.log 2>&1 || { grep error /tmp/mini5_host.log; exit 1; }
hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o /tmp/base5.s mini5.hip 2>/dev/null
for k in 0 1 2 3 4 5 6 7; do
  python3 - $k <<'PY'
import re, sys
k = int(sys.argv[1])
s = open('/tmp/base5.s').read()
m = re.search(r'\n(\.LBB0_\d+):[^\n]*\n(\ts_nop \d+\n)?\tv_mfma', s)          # the MFMA loop's head label (a hazard nop the compiler put behind it moves in front)
s = s.replace(m.group(0), '\n' + (m.group(2) or '') + '\ts_nop 0\n' * k + m.group(1) + ':\n\tv_mfma', 1)
open('/tmp/a5.s', 'w').write(s)
PY
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c /tmp/a5.s -o /tmp/a5.o && $LLVM/ld.lld -shared /tmp/a5.o -o /tmp/a5.co || { echo "k=$k: ASSEMBLY FAILED"; continue; }
  head_addr=$($LLVM/llvm-objdump -d --mcpu=gfx950 /tmp/a5.co 2>/dev/null | grep -m1 v_mfma | sed 's/.*\/\/ 0*\([0-9A-F]*\):.*/\1/')
  m=$(( 0x$head_addr % 32 ))
  if [ $m = 28 ] || [ $m = 0 ] || [ $m = 24 ]; then echo "first v_mfma at 0x$head_addr ($m mod 32):"; /tmp/mini5_host ${N:-20} 2000 3000 /tmp/a5.co; fi
done
