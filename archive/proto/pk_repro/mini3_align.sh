#!/bin/bash
# mini3.hip (explicit-register packed FMAs, self-checking against scalar FMAs) with its MFMA loop placed at 4 k bytes past its compiled
# position: does the compiler-independent form lose products once the loop head straddles a 32-byte boundary?
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}/archive/proto/pk_repro
LLVM=/opt/rocm/lib/llvm/bin
hipcc --offload-arch=gfx950 -O3 $KIND -o /tmp/mini3_host mini3.hip > /tmp/mini3_host.log 2>&1 || { grep error /tmp/mini3_host.log; exit 1; }
hipcc --offload-arch=gfx950 -O3 $KIND -S --cuda-device-only -o /tmp/base3.s mini3.hip 2>/dev/null
for k in 0 1 2 3 4 5 6 7 8; do
  python3 - $k <<'PY'
import sys
k = int(sys.argv[1])
s = open('/tmp/base3.s').read()
s = s.replace('.LBB0_3:', '\ts_nop 0\n' * k + '.LBB0_3:', 1)
open('/tmp/a3.s', 'w').write(s)
PY
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c /tmp/a3.s -o /tmp/a3.o && $LLVM/ld.lld -shared /tmp/a3.o -o /tmp/a3.co || { echo "k=$k: ASSEMBLY FAILED"; continue; }
  head_addr=$($LLVM/llvm-objdump -d --mcpu=gfx950 /tmp/a3.co 2>/dev/null | grep -m1 -A0 "${HEAD_PAT:-v_mfma}" | sed 's/.*\/\/ 0*\([0-9A-F]*\):.*/\1/')
  printf "first v_mfma at 0x%s (%2d mod 32): " $head_addr $(( 0x$head_addr % 32 ))
  /tmp/mini3_host ${N:-50} 2000 3000 /tmp/a3.co
done
