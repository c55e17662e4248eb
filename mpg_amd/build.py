"""Builds libmpg_hip.so (gfx950) in-tree with hipcc.  `python -m mpg_amd.build [--force]`.

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container; the resulting .so
travels with the repo snapshot to the GPU box (it is git-ignored, not gpurun-ignored)."""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'build')
LIB = os.path.join(HERE, 'libmpg_hip.so')
# the same sources with -DMPG_F32_MFMA: the exact-fp32 engine (v_mfma_f32_16x16x4_f32) beside the product - what bench.py
# times as `exact_fp32_ms_per_step` and the parity tests run as the second engine (mpg_amd/_lib.py ENGINES)
VARIANTS = {'split': (OBJ, LIB, []), 'f32': (os.path.join(HERE, 'build_f32'), os.path.join(HERE, 'libmpg_hip_f32.so'), ['-DMPG_F32_MFMA'])}
ARCH = os.environ.get('MPG_ARCH', 'gfx950')      # MPG_ARCH=gfx950:xnack- for experiments

# MPG_EXTRA_CFLAGS: ablation / diagnostic builds only (the experiment branches of archive/proto/ablation_macros.patch, -DMPG_STAMP, -DMPG_TIMELINE); never set in the product build
COMMON = os.environ.get('MPG_EXTRA_CFLAGS', '').split() + ['-O3', '-fPIC', '-std=c++17', '--offload-arch=' + ARCH, '-Wall', '-Wno-unused-function', '-Wno-bitwise-instead-of-logical',
          # a private array the optimiser cannot keep in registers must not be moved to LDS: the LDS copy is addressed by the
          # flattened thread id, whose workgroup size the code then reads from the dispatch packet in HOST memory
          # (microseconds per load; tools/dispatch_ptr_check.py, tests/test_abi.py keep every kernel free of such reads)
          '-mllvm', '-disable-promote-alloca-to-lds',
          # no SLP vectorizer anywhere: it is what forms v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 out of scalar float32 code, and
          # single products of such packed FMAs were lost nondeterministically beside matrix instructions ("lost
          # packed-FMA products"; reproducible at 99.5 % of launches with the packed-FMA experiment branch).  With it off no kernel that issues MFMAs
          # contains a packed fp32 instruction (tools/pk_census.py; tests/test_abi.py keeps it so).  Cost, same-box A/B of the bench
          # step (round 4): 0.2432 against 0.2400 / 0.2428 ms for the two baselines around it - inside the noise.
          '-fno-slp-vectorize',
          '-I' + os.path.join(HERE, '..', 'include')]
# per-file extras: the real-env kernel mirrors the reference op-by-op, so no fused multiply-adds there
EXTRA = {'env_path_tracking.hip': ['-ffp-contract=off'] + os.environ.get('MPG_ENV_CFLAGS', '').split(),
         # same reason of a different kind: the step and the fused step+store+reset kernel must round identically
         'env_cart_pole.hip': ['-ffp-contract=off'],
         # the two rollout sweeps are separate translation units so that each gets the scheduling options that suit it
         # (MPG_FWD_CFLAGS / MPG_BWD_CFLAGS override them in experiments: tools/ab.sh "FWD=... BWD=...")
         # forward sweep: max-memory-clause measures 64.5 us against 68.4 with the default strategy
         # (round 3: with the pendulum instantiations in their own translation unit - iterative-ilp crashes the register allocator on
         # them - iterative-ilp measures 63.7 - 63.9 us against 64.9 for max-memory-clause)
         'rollout_fwd.hip': os.environ.get('MPG_FWD_CFLAGS', '-mllvm -amdgpu-sched-strategy=iterative-ilp').split(),
         # experiments only (tools/ab_flags.sh): per-file flags of the other two engine translation units
         'fused_kernels.hip': os.environ.get('MPG_FUSED_CFLAGS', '').split(),
         # (the transposed-image experiment branch here - the transposed activation image for k_forward / k_backward only - measured k_forward -3.5 %, the TD3 step
         # at B = 65 536 0.896 -> 0.881 ms, null on the bench step (tools/ab_tr_files.sh, round 5); NOT shipped: its v_fma_mixhi_f16 forms
         # carry a low op_sel bit, which the containment rule of tests/test_abi.py keeps out of the shipped ISA altogether)
         'mlp_kernels.hip': os.environ.get('MPG_MLP_CFLAGS', '').split(),
         # reverse sweep: the same strategy measures 1.7 us faster than the default; without the SLP vectorizer (which turns the
         # model adjoint on the serial chain into v_pk_*_f32 plus the register moves that assemble their operand pairs) it
         # measures another 1.3 us faster (77.3 - 77.6 vs 78.7 - 78.9 us in alternating runs; no effect on the forward sweep)
         # and with the SLP vectorizer off the iterative-ilp strategy is the best of those that build (75.2 - 75.6 vs 76.6 - 77.5 us
         # for max-memory-clause, 80 default, 94 max-ilp; iterative-minreg / -maxocc: 2x slower or worse)
         'rollout_bwd.hip': os.environ.get('MPG_BWD_CFLAGS', '-mllvm -amdgpu-sched-strategy=iterative-ilp').split(),
         # the pendulum instantiations (NADP, config 3) measure 10 us slower under those and keep max-memory-clause with SLP
         # (MPG_BWDP_CFLAGS / MPG_FWDP_CFLAGS: experiments, tools/ab_side.sh)
         'rollout_bwd_pendulum.hip': os.environ.get('MPG_BWDP_CFLAGS', '-mllvm -amdgpu-sched-strategy=max-memory-clause').split(),
         'rollout_fwd_pendulum.hip': os.environ.get('MPG_FWDP_CFLAGS', '-mllvm -amdgpu-sched-strategy=max-memory-clause').split()}


def hipcc():
    for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found: libmpg_hip.so cannot be built (this package has no CPU fallback)')


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp')))


def _stamp(src, flags):
    h = hashlib.sha1(' '.join(flags).encode())
    for f in [src] + [os.path.join(CSRC, x) for x in sorted(os.listdir(CSRC)) if x.endswith('.h')] + \
            [os.path.join(HERE, '..', 'include', 'mpg_hip.h')]:
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force=False, verbose=True, jobs=None, engines=('split', 'f32')):
    """builds every engine variant (default: both); returns the product's path"""
    for e in engines:
        _build_one(e, force, verbose, jobs)
    return LIB


def _build_one(engine, force, verbose, jobs):
    from concurrent.futures import ThreadPoolExecutor
    OBJ, LIB, vflags = VARIANTS[engine]
    os.makedirs(OBJ, exist_ok=True)
    # objects (and their stamps / by-products) whose source no longer exists are removed: they would not be linked, but they travel
    # with every snapshot of the tree
    live = set(s + '.o' for s in sources())
    for f in os.listdir(OBJ):
        path = os.path.join(OBJ, f)
        if os.path.isfile(path) and (f.endswith('.o') or f.endswith('.o.stamp')) and not any(f == o or f.startswith(o + '.') for o in live):
            try:
                os.remove(path)
            except OSError:
                pass              # a concurrent build got there first
    cc = hipcc()
    objs, todo = [], []
    for s in sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s + '.o')
        flags = COMMON + vflags + EXTRA.get(s, []) + (['-x', 'hip'] if s.endswith('.hip') else [])
        stamp_file = obj + '.stamp'
        stamp = _stamp(src, flags)
        if force or not os.path.exists(obj) or not os.path.exists(stamp_file) or open(stamp_file).read() != stamp:
            todo.append(([cc] + flags + ['-c', src, '-o', obj], stamp_file, stamp))
        objs.append(obj)

    def compile_one(job):
        cmd, stamp_file, stamp = job
        if verbose:
            print('[mpg_amd.build]', ' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(stamp_file, 'w') as fh:
            fh.write(stamp)
    if todo:      # translation units are independent: compile them side by side
        with ThreadPoolExecutor(max_workers=jobs or min(len(todo), os.cpu_count() or 1, 16)) as ex:
            list(ex.map(compile_one, todo))
    if todo or not os.path.exists(LIB):
        cmd = [cc, '-shared', '-fPIC', '--offload-arch=' + ARCH, '-o', LIB] + objs
        if verbose:
            print('[mpg_amd.build]', ' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv, engines=('split',) if '--split-only' in sys.argv else ('split', 'f32'))
    print(LIB)
