// Row-per-lane ("CT") form of the weight-stationary 2x256 ELU MLP engine (gfx950), used by the two rollout sweeps.
//
// mlp_core.h issues every hidden-layer product as  D[row][unit] = A[row][k] * B[k][unit]  with the activations as the MFMA's
// A operand: a lane then owns FOUR rows x ONE unit per tile ("C layout").  Here the SAME instruction gets its operands
// swapped - the stationary weights are A, the activations B - which yields the transposed tile D^T[unit][row]: lane
// (row = l & 15, rg = l >> 4) owns ONE batch row and, over the wave's two tiles, the EIGHT units 32 kb + 8 kg + (0..7) of one
// 16-byte chunk (kb, kg) of the split-fp16 LDS image (unit_of() in mlp_core.h is chosen so that both engines read the same
// packed weight images).  What that buys per policy evaluation and wave, against the C layout:
//   * the activation image is written with ONE ds_write_b128 per lane and image (hi / lo) instead of eight ds_write_b32
//     behind eight DPP exchanges + eight selects, and the fp16 split is four v_fma_mix per pair instead of seven instructions;
//   * layer 1 stays on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, operands swapped like the rest; 4 k-steps for up to 16
//     inputs).  A split-fp16 form - one v_mfma_f32_16x16x32_f16 per tile with the hi and lo halves of inputs and weights
//     sharing its K = 32 - was built and measured: the fp16 split of the next input then sits on the trajectory lanes' serial
//     chain and costs more than the two 32-cycle MFMAs it saves (forward sweep 70.1 vs 66.2 us);
//   * the output layer is 8 fmas per output + a two-stage v_permlane swap reduction (12 instructions for two outputs)
//     instead of 16 multiplies + 32 DPP adds;
//   * the reverse sweep's input gradient dx = dz1 W1^T takes its B operand straight from the registers that hold dz1
//     (no float32 LDS image, no eight 32-cycle fp32 MFMAs): three f16 MFMAs;
//   * per-row quantities (dL/dz3, the row exponent of the reverse layer) are per-LANE scalars.
// Activations handed to the weight-gradient kernel keep mlp_core.h's G16 layout: the (few) steps that stash for it
// transpose through a wave-private LDS tile (g16_store / g16_load).
#pragma once
#include "mlp_core.h"

namespace mlp {
namespace ct {

struct LaneCT {
    int lane, wave, row, rg, kb, kg;
    __device__ LaneCT() {
        lane = threadIdx.x & 63;
        wave = threadIdx.x >> 6;
        row = lane & 15;
        rg = lane >> 4;
        kb = 2 * (wave & 3) + (rg & 1);       // = unit_of(wave, t, 4 rg + j) >> 5
        kg = 2 * (wave >> 2) + (rg >> 1);     // = (unit_of(..) >> 3) & 3
    }
    // the lane's p-th value (p = 4 t + j: tile t, accumulator register j) is hidden unit 32 kb + 8 kg + p
    __device__ int unit(int p) const { return 32 * kb + 8 * kg + p; }
};

constexpr float UNSCALE = 1.f / (W_SCALE * A_SCALE);

// (a, b) * scale -> packed fp16 pairs  hi = (f16(a s), f16(b s)),  lo = (f16(a s - hi.x), f16(b s - hi.y)).
// scale is a power of two (exact); four v_fma_mix instead of the seven instructions the C expression compiles to.
__device__ __forceinline__ void split_pair(float a, float b, float scale, float& hi, float& lo) {
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(a), "v"(scale));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(b), "v"(scale));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(a), "v"(scale), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(b), "v"(scale), "v"(hi));
}

// the lane's 8 values (its chunk of the row) -> hi / lo fragments (8 halves each)
__device__ __forceinline__ void split8(const float (&v)[2][4], float scale, f32x4& hi, f32x4& lo) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float h, l;
            split_pair(v[t][2 * u], v[t][2 * u + 1], scale, h, l);
            hi[2 * t + u] = h;
            lo[2 * t + u] = l;
        }
}

// the lane's chunk of both images: one 16-byte store each (8 x 8 contiguous lanes per LDS pass, 32 distinct banks)
__device__ __forceinline__ void store_image(float* sA, const LaneCT& L, const float (&v)[2][4], float scale) {
    f32x4 hi, lo;
    split8(v, scale, hi, lo);
    _Float16* p = reinterpret_cast<_Float16*>(sA) + L.kg * PLANE_H + L.row * ROW_H + 8 * L.kb;
    *reinterpret_cast<f32x4*>(p) = hi;
    *reinterpret_cast<f32x4*>(p + IMG_H) = lo;
}

__device__ __forceinline__ f16x8 wfrag(const float (&w)[128], int v) {
    return __builtin_bit_cast(f16x8, f32x4{w[4 * v], w[4 * v + 1], w[4 * v + 2], w[4 * v + 3]});
}

// D^T = W (stationary, A operand) x activations (LDS images hi / lo, B operand): for the lane's row, the products of its 8
// units with all 256 contraction units.  m0 / m1: tile 0 / 1, in units of W_SCALE * (the image's scale).
__device__ __forceinline__ void mm256(const float* sA, const LaneCT& L, const float (&w)[128], f32x4& m0, f32x4& m1) {
    const _Float16* bh = reinterpret_cast<const _Float16*>(sA) + L.rg * PLANE_H + L.row * ROW_H;   // B[k = 8 (l>>4) + ..][n = l & 15]
    const _Float16* bl = bh + IMG_H;
#ifdef MPG_AB_NOMFMA
    constexpr int NKB = 1;
#else
    constexpr int NKB = 8;
#endif
    m0 = f32x4{0.f, 0.f, 0.f, 0.f};
    m1 = m0;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const f16x8 ah = *reinterpret_cast<const f16x8*>(bh + 8 * kb), al = *reinterpret_cast<const f16x8*>(bl + 8 * kb);
        const int v0 = (kb * 2 + 0) * 2, v1 = (kb * 2 + 1) * 2;
        m0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfrag(w, v0), ah, m0, 0, 0, 0);
        m1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfrag(w, v1), ah, m1, 0, 0, 0);
        m0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfrag(w, v0 + 1), ah, m0, 0, 0, 0);
        m1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfrag(w, v1 + 1), ah, m1, 0, 0, 0);
        m0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfrag(w, v0), al, m0, 0, 0, 0);
        m1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfrag(w, v1), al, m1, 0, 0, 0);
    }
}

// ---- network input block: plain float32 [16 rows][8 INB], one lane per row writes it ----------------------------------------
template <int INB>
__device__ __forceinline__ void store_x_block(float* sX, int row, const float (&x)[8 * INB]) {
#pragma unroll
    for (int q = 0; q < 2 * INB; ++q)
        *reinterpret_cast<f32x4*>(sX + row * (8 * INB) + 4 * q) = f32x4{x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
}

// ---- small per-lane stationary pieces ------------------------------------------------------------------------------
// INB: input chunks of 8 (1: up to 8 inputs, 2: up to 16).  FWD / BWD select what a sweep needs.
template <int INB, int OU>
struct SmallCT {
    float w1f[2 * INB][2]; // layer 1: A operand of v_mfma_f32_16x16x4_f32, k-step q, tile t: lane (m = l&15, g = l>>4) holds W1[4 q + g][unit_of(w, t, m)]
    float b1[2][4], b2[2][4];
    float w3[2][4][OU];    // W3[unit(p)][o]
};

template <int INB, int OU>
__device__ __forceinline__ void load_small_fwd(const Net& n, int in_dim, const LaneCT& L, SmallCT<INB, OU>& r) {
    const int m = L.lane & 15, g = L.lane >> 4;
#pragma unroll
    for (int q = 0; q < 2 * INB; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t) r.w1f[q][t] = (4 * q + g) < in_dim ? n.W1[(4 * q + g) * H + unit_of(L.wave, t, m)] : 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = L.unit(4 * t + j);
            r.b1[t][j] = n.b1[u];
            r.b2[t][j] = n.b2[u];
#pragma unroll
            for (int o = 0; o < OU; ++o) r.w3[t][j][o] = n.W3[u * n.out_dim + o];
        }
}

template <int OU>
struct SmallCTB {
    f32x4 w1h, w1l;        // dx product A operand: lane (m = input i = l&15, g = l>>4) holds W1[i][unit_of(w, t, 4 g + j)] * 64 over p = 4 t + j
    f32x4 w1m;             // third part of a three-way split (hi + lo + this): W1 then enters with all 24 bits
};

template <int OU>
__device__ __forceinline__ void load_small_bwd(const Net& n, int in_dim, const LaneCT& L, SmallCTB<OU>& r) {
    const int i = L.lane & 15, g = L.lane >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {          // pairs p = 2q, 2q + 1: tile t = q >> 1, registers j = 2 (q & 1), + 1
        const int u0 = unit_of(L.wave, q >> 1, 4 * g + 2 * (q & 1));
        const float a = i < in_dim ? n.W1[i * H + u0] : 0.f, b = i < in_dim ? n.W1[i * H + u0 + 1] : 0.f;
        float hi, lo;
        split_pair(a, b, W_SCALE, hi, lo);
        r.w1h[q] = hi;
        r.w1l[q] = lo;
        const f16x2 h2 = __builtin_bit_cast(f16x2, hi), l2 = __builtin_bit_cast(f16x2, lo);
        const f16x2 m2 = {(_Float16)((a * W_SCALE - (float)h2[0]) - (float)l2[0]), (_Float16)((b * W_SCALE - (float)h2[1]) - (float)l2[1])};
        r.w1m[q] = __builtin_bit_cast(float, m2);
    }
}

// The reverse sweep keeps W3 in LDS, not in registers (it has none to spare: the compiler otherwise spills part of the stationary
// image and reloads it every step): sW3 [256 units][OU], a lane's 8 units are 8 OU consecutive floats.
template <int OU>
__device__ __forceinline__ void stage_w3(const Net& n, float* sW3) {
    for (int i = threadIdx.x; i < H * OU; i += NTHREAD) sW3[i] = n.W3[(i / OU) * n.out_dim + (i % OU)];
}

// ---- lane-private stash (forward sweep -> reverse sweep): float4 index ((group*16 + 2 wave + t)*64 + lane) -----------
__device__ __forceinline__ void stash_store(float* __restrict__ base, long group, const LaneCT& L, const float (&v)[2][4]) {
    f32x4* p = reinterpret_cast<f32x4*>(base) + (group * 16 + 2 * L.wave) * 64 + L.lane;
    p[0] = f32x4{v[0][0], v[0][1], v[0][2], v[0][3]};
    p[64] = f32x4{v[1][0], v[1][1], v[1][2], v[1][3]};
}
__device__ __forceinline__ void stash_load(const float* __restrict__ base, long group, const LaneCT& L, float (&v)[2][4]) {
    const f32x4* p = reinterpret_cast<const f32x4*>(base) + (group * 16 + 2 * L.wave) * 64 + L.lane;
    const f32x4 a = p[0], b = p[64];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[0][j] = a[j];
        v[1][j] = b[j];
    }
}

// ---- G16 stash (what the weight-gradient kernel reads): transpose through a wave-private LDS tile ---------------------
// sT: this wave's 2 x 16 x T_LD floats.  The CT lane holds (row, slots 4 rg + j) of tile t; G16 lane (c = slot, rq = row quad)
// wants rows 4 rq .. 4 rq + 3 of slot c.
constexpr int T_LD = 20;
constexpr int T_FLOATS = 2 * GROUP * T_LD;          // per wave
__device__ __forceinline__ void g16_store(float* __restrict__ base, long group, const LaneCT& L, float* sT, const float (&v)[2][4]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) sT[(t * GROUP + 4 * L.rg + j) * T_LD + L.row] = v[t][j];
    __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): the wave reads back what it wrote itself
    __builtin_amdgcn_wave_barrier();
    f32x4* p = reinterpret_cast<f32x4*>(base) + (group * 16 + 2 * L.wave) * 64 + L.lane;
#pragma unroll
    for (int t = 0; t < 2; ++t) p[64 * t] = *reinterpret_cast<const f32x4*>(sT + (t * GROUP + (L.lane & 15)) * T_LD + 4 * (L.lane >> 4));
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();             // the tile may be rewritten right away
}
__device__ __forceinline__ void g16_load(const float* __restrict__ base, long group, const LaneCT& L, float* sT, float (&v)[2][4]) {
    const f32x4* p = reinterpret_cast<const f32x4*>(base) + (group * 16 + 2 * L.wave) * 64 + L.lane;
#pragma unroll
    for (int t = 0; t < 2; ++t) *reinterpret_cast<f32x4*>(sT + (t * GROUP + (L.lane & 15)) * T_LD + 4 * (L.lane >> 4)) = p[64 * t];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[t][j] = sT[(t * GROUP + 4 * L.rg + j) * T_LD + L.row];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
}

// ---- sum over the four lanes that share a row (rg = 0..3): two swap stages, every lane ends with the total -------------
__device__ __forceinline__ float sum_over_rg(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// ---- layer 1: pre-activations of the lane's 8 units from the input block (exact fp32) --------------------------------------------
template <int INB, int OU>
__device__ __forceinline__ void layer1(const float* sXi, const LaneCT& L, const SmallCT<INB, OU>& r, float (&h1)[2][4]) {
    // B operand of k-step q: x[row][4 q + g]
    f32x4 z0 = {r.b1[0][0], r.b1[0][1], r.b1[0][2], r.b1[0][3]}, z1 = {r.b1[1][0], r.b1[1][1], r.b1[1][2], r.b1[1][3]};
#pragma unroll
    for (int q = 0; q < 2 * INB; ++q) {
        const float x = sXi[L.row * (8 * INB) + 4 * q + L.rg];
        z0 = __builtin_amdgcn_mfma_f32_16x16x4f32(r.w1f[q][0], x, z0, 0, 0, 0);
        z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(r.w1f[q][1], x, z1, 0, 0, 0);
    }
    elu8(z0, z1, h1);
}

// ---- forward through both hidden layers + output partials for the workgroup's row group ---------------------------------
// sXi: input block (store_x_block, published before the call); sA: activation images; sPart [NWAVE][16][MAXOUT]: per-wave partial sums of
// h2 W3 (no bias), written by the lanes with rg == 0.  Ends BEFORE the barrier that publishes sPart.
// h1_stash (nullable): h1 leaves for the stash right behind its image store, so that the store drains under the MFMA block
// (lane-private layout, or G16 through the wave's transpose tile sT when h1_g16).
template <int INB, int OU>
__device__ __forceinline__ void forward_ct(const float* sXi, float* sA, float* sPart, const LaneCT& L, const float (&w2)[128],
                                           const SmallCT<INB, OU>& r, float (&h1)[2][4], float (&h2)[2][4],
                                           float* h1_stash = nullptr, long stash_group = 0, bool h1_g16 = false, float* sT = nullptr) {
    layer1<INB, OU>(sXi, L, r, h1);
    store_image(sA, L, h1, A_SCALE);
    if (h1_stash) {
        if (h1_g16) g16_store(h1_stash, stash_group, L, sT, h1);
        else stash_store(h1_stash, stash_group, L, h1);
    }
    MPG_STAMP_AT(1);
    lds_barrier();
    MPG_STAMP_AT(2);
    f32x4 m0, m1;
    mm256(sA, L, w2, m0, m1);
    MPG_STAMP_AT(3);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m0[j] = fmaf(m0[j], UNSCALE, r.b2[0][j]);
        m1[j] = fmaf(m1[j], UNSCALE, r.b2[1][j]);
    }
    elu8(m0, m1, h2);
    float p[OU];
#pragma unroll
    for (int o = 0; o < OU; ++o) {
        float s0 = h2[0][0] * r.w3[0][0][o], s1 = h2[1][0] * r.w3[1][0][o];
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            s0 = fmaf(h2[0][j], r.w3[0][j][o], s0);
            s1 = fmaf(h2[1][j], r.w3[1][j][o], s1);
        }
        p[o] = sum_over_rg(s0 + s1);
    }
    if (L.rg == 0) {
#pragma unroll
        for (int o = 0; o < OU; ++o) sPart[(L.wave * GROUP + L.row) * MAXOUT + o] = p[o];
    }
    MPG_STAMP_AT(4);
}

// ---- reverse pass, first half: dz2 = (dz3 W3^T) * ELU'(h2) into the images, scaled by 2^-e of the lane's row ------------------
// sD3 [MAXOUT][16] (d3_index): dL/dz3 of the used outputs.  Returns the row exponent e (dz2 entered the image as dz2 * 2^(4 - e)).
template <int OU>
__device__ __forceinline__ int backward_dz2_ct(const float* sD3, float* sA, const LaneCT& L, const float* sW3,
                                               const float (&h2)[2][4], float (&dz2)[2][4]) {
    float w3[2][4][OU];
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(sW3 + L.unit(0) * OU);
#pragma unroll
        for (int q = 0; q < 2 * OU; ++q) {
            const f32x4 v = wp[q];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = 4 * q + i;                  // float f of the lane's 8 * OU: unit p = f / OU, output f % OU
                w3[(f / OU) >> 2][(f / OU) & 3][f % OU] = v[i];
            }
        }
    }
    float d3[OU], mx = 0.f;
#pragma unroll
    for (int o = 0; o < OU; ++o) {
        d3[o] = sD3[d3_index(L.row, o)];
        mx = fmaxf(mx, fabsf(d3[o]));
    }
    int e = __builtin_amdgcn_frexp_expf(mx);
    e = e < -100 ? -100 : e;                      // 2^(4 - e) must stay finite; rows that small contribute nothing in float32 anyway
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float dh = d3[0] * w3[t][j][0];
#pragma unroll
            for (int o = 1; o < OU; ++o) dh = fmaf(d3[o], w3[t][j][o], dh);
            dz2[t][j] = dh * elu_grad_from_out(h2[t][j]);
        }
    store_image(sA, L, dz2, ldexpf(A_SCALE, -e));
    return e;
}

// ---- reverse pass, second half: dh1 = dz2 W2^T, dz1 = dh1 * ELU'(h1) (in units of 2^e), input-gradient partials ----------------
// dz1s: dz1 * 2^-e.  If WANT_DX the wave's partial sums of dz1 W1^T go to sPartX [NWAVE][16][XS] (XS >= 4 * ceil(IN / 4)).
template <int OU, int XSW, bool WANT_DX>
__device__ __forceinline__ void backward_rest_ct(const float* sA, float* sPartX, const LaneCT& L, const float (&w2t)[128],
                                                 const SmallCTB<OU>& r, const float (&h1)[2][4], int e, int in_dim, float (&dz1s)[2][4]) {
    f32x4 m0, m1;
    mm256(sA, L, w2t, m0, m1);
    MPG_STAMP_AT(3);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        dz1s[0][j] = (m0[j] * UNSCALE) * elu_grad_from_out(h1[0][j]);
        dz1s[1][j] = (m1[j] * UNSCALE) * elu_grad_from_out(h1[1][j]);
    }
    if (WANT_DX) {
        // dx^T[i][row] = sum over the wave's 32 units of W1[i][unit] dz1[row][unit]: the lane's 8 values ARE its B fragment
        f32x4 bh, bl;
        split8(dz1s, A_SCALE, bh, bl);
        f32x4 dx = {0.f, 0.f, 0.f, 0.f};
        const f16x8 ah = __builtin_bit_cast(f16x8, r.w1h), al = __builtin_bit_cast(f16x8, r.w1l);
        dx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, __builtin_bit_cast(f16x8, bh), dx, 0, 0, 0);
        dx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, __builtin_bit_cast(f16x8, bh), dx, 0, 0, 0);
        dx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, __builtin_bit_cast(f16x8, bl), dx, 0, 0, 0);
        dx = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, r.w1m), __builtin_bit_cast(f16x8, bh), dx, 0, 0, 0);
        if (4 * L.rg < in_dim) {        // lane (rg, row) holds dx[row][4 rg .. 4 rg + 3]
            const float un = ldexpf(UNSCALE, e);
            *reinterpret_cast<f32x4*>(sPartX + (L.wave * GROUP + L.row) * XSW + 4 * L.rg) = f32x4{dx[0] * un, dx[1] * un, dx[2] * un, dx[3] * un};
        }
    }
    MPG_STAMP_AT(4);
}

}  // namespace ct
}  // namespace mlp
