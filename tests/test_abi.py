"""CPU-side checks of the drop-in boundary: the library builds for gfx950, loads, and exports every
symbol include/mpg_hip.h declares.  No compute call is made (no GPU here)."""
import ctypes
import os
import subprocess

import pytest

import mpg_amd._lib as L
from mpg_amd import build as B


@pytest.fixture(scope='module')
def built():
    return B.build(verbose=False)


def test_library_builds_and_exports_every_declared_symbol(built):
    assert os.path.exists(built)
    names = L.declared_symbols()
    assert 'mpg_env_step' in names and len(names) >= 5
    lib = ctypes.CDLL(built)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.mpg_abi_version() == 10         # mpg_common.h: bumped with every layout or signature change (ops.py mirrors the structs)


def test_no_undeclared_exports(built):
    out = subprocess.check_output(['nm', '-D', '--defined-only', built]).decode()
    exported = sorted(l.split()[-1] for l in out.splitlines() if ' T mpg_' in l)
    assert exported == L.declared_symbols()


def test_code_object_is_gfx950_only(built):
    import re
    blob = open(built, 'rb').read()
    targets = set(re.findall(rb'amdgcn-amd-amdhsa--(gfx[0-9a-z]+)', blob))
    assert targets == {b'gfx950'}, targets


def test_product_path_fails_loudly_without_library(monkeypatch, tmp_path):
    monkeypatch.setattr(L, '_libs', {})
    monkeypatch.setattr(L, 'ENGINES', {'split': str(tmp_path / 'nope.so'), 'f32': str(tmp_path / 'nope_f32.so')})
    with pytest.raises(L.MpgError):
        L.lib()


def test_exact_fp32_engine_is_built_beside_the_product_with_the_same_abi(built):
    """mpg_amd/build.py builds the library twice from the same sources: the product and libmpg_hip_f32.so (-DMPG_F32_MFMA, the
    exact-fp32 hidden-layer engine that bench.py times as exact_fp32_ms_per_step and the parity tests run as the second engine).
    Same exports, same ABI version; only the product contains f16 matrix instructions."""
    f32 = L.ENGINES['f32']
    assert os.path.exists(f32) and os.path.dirname(f32) == os.path.dirname(built)
    lib = ctypes.CDLL(f32)
    assert not [n for n in L.declared_symbols() if not hasattr(lib, n)]
    assert lib.mpg_abi_version() == ctypes.CDLL(built).mpg_abi_version()


def test_product_package_never_imports_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dp, _, fs in os.walk(os.path.join(root, 'mpg_amd')):
        for f in fs:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f


def test_no_kernel_reads_the_dispatch_or_queue_packet():
    """The AQL dispatch packet and the queue descriptor live in host memory; a kernel whose descriptor enables their
    pointer reads them with scalar loads that cost microseconds (round 2: `p.v[i & 3]` on a Philox draw made the compiler
    move the four words to LDS and fetch the workgroup size from the packet - 7 to 30 us on the lanes that draw the
    minibatch).  Cross-compiles every translation unit and inspects the kernel descriptors (no GPU needed)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('dispatch_ptr_check', os.path.join(os.path.dirname(__file__), '..', 'tools',
                                                                                    'dispatch_ptr_check.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.offenders() == []


def test_no_dpp_instruction_reads_a_register_inside_its_hazard_window():
    """A DPP instruction needs two wait states behind the VALU write of its DPP operand; the assembler does not check inline
    asm, and the engine's output reduction is hand-written `v_add_f32_dpp` stages (mlp_core.h) that rely on their stage-major
    order.  tools/dpp_hazard_check.py verifies the property on the ISA of every translation unit as the shipped flags compile
    it; the checker itself is first shown to fire on a synthetic violation."""
    import os
    import sys
    tools = os.path.join(os.path.dirname(__file__), '..', 'tools')
    sys.path.insert(0, tools)
    try:
        import dpp_hazard_check as C
    finally:
        sys.path.remove(tools)
    bad_asm = """
_Zfoo:
    v_fma_f32 v3, v1, v2, v3
    v_add_f32_dpp v4, v3, v3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf
"""
    ok_asm = """
_Zfoo:
    v_fma_f32 v3, v1, v2, v3
    s_nop 1
    v_add_f32_dpp v4, v3, v3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf
    v_fma_f32 v[8:11], v1, v2, v3
    v_mov_b32_e32 v20, v21
    v_mov_b32_dpp v5, v9 row_mirror row_mask:0xf bank_mask:0xf
"""
    b, n = C.violations(bad_asm)
    assert n == 1 and len(b) == 1
    b, n = C.violations(ok_asm)
    assert n == 2 and len(b) == 1 and 'v_mov_b32_dpp' in b[0][2]          # one wait state is not enough, a range write counts
    bad, n = C.check()
    assert n > 1000 and bad == [], bad[:5]


def test_library_sources_read_no_environment_variables():
    """DESIGN.md section 1, "No process-wide state": the library's behaviour is a function of its arguments - no getenv in the
    product sources (A/B variants are compile-time macros, tools/ab.sh)."""
    import os
    import re
    csrc = os.path.join(os.path.dirname(__file__), '..', 'mpg_amd', 'csrc')
    for f in sorted(os.listdir(csrc)):
        if f.endswith(('.hip', '.h', '.cpp')):
            assert not re.search(r'\bgetenv\s*\(', open(os.path.join(csrc, f)).read()), f


def test_no_packed_fp32_arithmetic_beside_matrix_instructions():
    """Containment of the lost packed-FMA products BY CONSTRUCTION: in the round-2 weight-gradient
    kernel single products of v_pk_fma_f32 were lost nondeterministically (99.5 % of launches with the explicit packed form,
    tools/pk_anomaly.sh), only in a kernel that also issues MFMAs.  The library is therefore built without the SLP vectorizer
    (mpg_amd/build.py COMMON) and tools/pk_census.py lists, from the ISA of every translation unit as the shipped flags compile
    it, the kernels that contain v_pk_{fma,mul,add}_f32: none of them may contain a matrix instruction."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('pk_census', os.path.join(os.path.dirname(__file__), '..', 'tools', 'pk_census.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rows = mod.census()
    beside = [(f, mod.demangle(k), c) for f, k, c, mf in rows if mf]
    assert beside == [], beside
    # round 4 (archive/proto/pk_repro): the victims are packed fp32 operations whose LOW lane takes the HIGH register of src1
    # (op_sel[1] set), beside a v_mfma_f32_16x16x32_f16 loop of ANY co-resident kernel whose head straddles a 32-byte boundary.
    # The shipped ISA has no instruction with a low op_sel bit at all (only the op_sel_hi forms of v_fma_mixlo/hi_f16): keep it so.
    assert mod.OPSEL_FOUND == [], mod.OPSEL_FOUND[:10]


def test_shipped_sources_hold_no_experiment_scaffolding():
    """the shipped kernels carry the shipped expansion only - no conditional on an ablation / layout experiment
    macro is left in mpg_amd/csrc (they live in archive/proto/ablation_macros.patch; tools/strip_ablation.py removed them with the
    device and host assembly of every translation unit unchanged, tools/isa_hash.py)."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'strip_ablation.py'), '--check'], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    pat = re.compile(r'MPG_AB_|MPG_IMG_LAYOUT|MPG_TR_IMAGE|MPG_IMG_INTERLEAVE')
    csrc = os.path.join(root, 'mpg_amd', 'csrc')
    hits = [f for f in sorted(os.listdir(csrc)) if pat.search(open(os.path.join(csrc, f)).read())]
    assert not hits, hits
    assert os.path.exists(os.path.join(root, 'archive', 'proto', 'ablation_macros.patch'))
