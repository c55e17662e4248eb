#!/bin/bash
# Instruction-level experiments on the compiler's own assembly of mini2.hip (-DK_EXEC=0 -DSTEPS=1): the .s is edited by the python
# snippets below, assembled into a code object and launched by mini2's host code (argv[4]).   bash archive/proto/pk_repro/asm_bisect.sh
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/../../..}/archive/proto/pk_repro
LLVM=/opt/rocm/lib/llvm/bin
F="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DK_EXEC=0 -DSTEPS=1"
hipcc $F -o /tmp/mini2_host mini2.hip > /tmp/mini2_host.log 2>&1 || { grep error /tmp/mini2_host.log; exit 1; }
hipcc $F -S --cuda-device-only -o /tmp/base.s mini2.hip 2>/dev/null
run() {   # name, python edit of the text `s`
  python3 - "$1" <<PY
import re, sys
s = open('/tmp/base.s').read()
$2
open('/tmp/v_' + sys.argv[1] + '.s', 'w').write(s)
PY
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c /tmp/v_$1.s -o /tmp/v_$1.o 2>/tmp/v_$1.log && $LLVM/ld.lld -shared /tmp/v_$1.o -o /tmp/v_$1.co || { echo "$1: ASSEMBLY FAILED"; head -3 /tmp/v_$1.log; return; }
  printf "%-28s " "$1"; /tmp/mini2_host ${N:-100} 600 0 /tmp/v_$1.co | head -${LINES_SHOWN:-1} | cut -c1-200
}
run base "pass"
run shift4 "s = s.replace('_Z7k_mini2PKfS0_iPf:', '_Z7k_mini2PKfS0_iPf:\\n\\ts_nop 0', 1)"
run shift8 "s = s.replace('_Z7k_mini2PKfS0_iPf:', '_Z7k_mini2PKfS0_iPf:\\n\\ts_nop 0\\n\\ts_nop 0', 1)"
run nop_before_opsel "s = re.sub(r'(\\tv_pk_fma_f32 [^\\n]*op_sel:\\[0,1,0\\])', r'\\ts_nop 7\\n\\1', s)"
run nop_after_opsel "s = re.sub(r'(\\tv_pk_fma_f32 [^\\n]*op_sel:\\[0,1,0\\][^\\n]*\\n)', r'\\1\\ts_nop 7\\n', s)"
run nop_after_every_pk "s = re.sub(r'(\\tv_pk_fma_f32 [^\\n]*\\n)', r'\\1\\ts_nop 1\\n', s)"
run lds_all_landed "s = s.replace('s_waitcnt lgkmcnt(7)', 's_waitcnt lgkmcnt(0)\\n\\ts_nop 7', 1)"
run opsel_via_copy "s = re.sub(r'\\tv_pk_fma_f32 (v\\[\\d+:\\d+\\]), (v\\[\\d+:\\d+\\]), v\\[(\\d+):(\\d+)\\], (v\\[\\d+:\\d+\\]) op_sel:\\[0,1,0\\]', lambda m: '\\tv_mov_b32 v60, v%s\\n\\tv_pk_fma_f32 %s, %s, v[60:61], %s op_sel_hi:[1,0,1]' % (m.group(4), m.group(1), m.group(2), m.group(5)), s); s = re.sub(r'\.amdhsa_next_free_vgpr \d+', '.amdhsa_next_free_vgpr 64', s)"
# whose placement matters: the packed-FMA code's (label .LBB0_10 on) or the MFMA loop's (in front of it)?
for n in 1 2 4 8 16; do
  run thin_shift_$((4*n)) "s = s.replace('.LBB0_10:', '.LBB0_10:' + '\\n\\ts_nop 0' * $n, 1)"
done
run mfma_shift4_thin_plus64 "s = s.replace('_Z7k_mini2PKfS0_iPf:', '_Z7k_mini2PKfS0_iPf:\\n\\ts_nop 0', 1).replace('.LBB0_10:', '\\ts_nop 0' + '\\n\\ts_nop 0' * 14 + '\\n.LBB0_10:', 1)"
run both_shift64 "s = s.replace('_Z7k_mini2PKfS0_iPf:', '_Z7k_mini2PKfS0_iPf:' + '\\n\\ts_nop 0' * 16, 1)"
