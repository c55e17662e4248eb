#!/bin/bash
# The MFMA loop of the NEIGHBOUR workgroups shifted by 4 k bytes (k = 0 .. 16), the packed-FMA code kept at its alignment (+64 bytes):
# which placements of the matrix loop make the other waves lose products?   bash archive/proto/pk_repro/asm_align.sh
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}/archive/proto/pk_repro
LLVM=/opt/rocm/lib/llvm/bin
F="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DK_EXEC=0 -DSTEPS=1"
hipcc $F -o /tmp/mini2_host mini2.hip > /tmp/mini2_host.log 2>&1 || { grep error /tmp/mini2_host.log; exit 1; }
hipcc $F -S --cuda-device-only -o /tmp/base.s mini2.hip 2>/dev/null
for k in 0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do
  python3 - $k <<'PY'
import sys
k = int(sys.argv[1])
s = open('/tmp/base.s').read()
s = s.replace('_Z7k_mini2PKfS0_iPf:', '_Z7k_mini2PKfS0_iPf:' + '\n\ts_nop 0' * k, 1)
s = s.replace('.LBB0_10:', '\ts_nop 0\n' * (16 - k) + '.LBB0_10:', 1)
open('/tmp/a.s', 'w').write(s)
PY
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c /tmp/a.s -o /tmp/a.o && $LLVM/ld.lld -shared /tmp/a.o -o /tmp/a.co || { echo "k=$k: ASSEMBLY FAILED"; continue; }
  head_addr=$($LLVM/llvm-objdump -d --mcpu=gfx950 /tmp/a.co | grep -m1 v_mfma | sed 's/.*\/\/ 0*\([0-9A-F]*\):.*/\1/')
  printf "matrix loop shifted by %2d bytes (first v_mfma at 0x%s, %2d mod 64): " $((4*k)) $head_addr $(( 0x$head_addr % 64 ))
  /tmp/mini2_host ${N:-60} 600 0 /tmp/a.co | head -1 | sed 's/.*STEPS=1: //' | cut -c1-110
done
