// Weight-stationary 2x256 ELU MLP engine for gfx950 (device code shared by every network kernel).
//
// Geometry (fixed): one workgroup = 512 threads = 8 waves = 2 waves per SIMD.  A workgroup works on one
// ROW GROUP of 16 rows at a time.  Wave w owns hidden columns [32w, 32w+32) = 2 column tiles of 16.
// The 256x256 hidden kernel W2 lives in REGISTERS for the whole kernel: each wave keeps its 256x32 slice as
// 128 MFMA B-operands (v_mfma_f32_16x16x4_f32: exact fp32, k-ordered fma chain), so the hot loop touches
// neither HBM nor L2 for weights; activations reach the MFMA A-operand through a 16.6 KB LDS image.
//
// Ownership of a 16x256 activation block ("C layout", also the MFMA C/D layout): lane l of wave w holds, for
// tile t in {0,1} and j in 0..3, element (row = 4*(l>>4) + j, col = 32*w + 16*t + (l&15)).
//
// "G16" global layout for stashed activations: float4 index ((group*16 + col/16)*64 + lane), the 4 floats
// being j = 0..3.  Every lane re-reads exactly the float4 it wrote: 1 KiB coalesced per wave instruction,
// and the same float4 is directly the A / B fragment of the weight-gradient MFMA (k = batch row).
#pragma once
#include "mpg_common.h"

namespace mlp {

constexpr int H = 256;
constexpr int NTHREAD = 512;
constexpr int NWAVE = 8;
constexpr int GROUP = 16;      // rows per row group
constexpr int A_IMG = 4608;    // floats reserved per LDS A image region: >= GROUP*LDA (float32 image) and = 2 split-fp16 images
constexpr int LDA = 280;       // row stride (floats) of the float32 LDS A image
constexpr int KS = 68;         // stride between the 4 k-phases of a row; (LDA, KS) = (280, 68): b128 reads conflict-free, b32 writes 2-way (free)
constexpr int XS = 8;          // row stride of the small LDS input block for networks with up to 8 inputs
// Networks with 9 .. 16 inputs (observations with look-ahead terms, path_tracking_env.py:385-402: obs_dim = 6 + num_future_data;
// SURVEY f3) run the same engine with a 16-wide input block: template parameter IN = 16 stands for "up to 16, the actual width
// is Net::in_dim", layer 1 is four k-steps of the fp32 MFMA instead of two, the dx partials are 16 wide.  IN = 24: up to 24
// inputs (the critics of num_future_data = 9, 10: 17 / 18 inputs) - six k-steps, a second 16-column tile of dx partials.
template <int IN>
__host__ __device__ constexpr int xs_of() { return IN <= 8 ? 8 : (IN <= 16 ? 16 : 24); }
template <int IN>
__host__ __device__ constexpr int l1_steps() { return IN <= 8 ? 2 : (IN <= 16 ? 4 : 6); }
constexpr int MAXOUT = 2;      // outputs ever *used* (policy mean: act_dim <= 2; critic: 1)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- arithmetic of the 256 x 256 hidden layer -------------------------------------------------------------------
// Default: SPLIT-FP16.  Every float32 operand x of the hidden-layer product is written as hi + lo with hi = fp16(x),
// lo = fp16(x - hi) - 22..23 significand bits, exact products - and the product is evaluated as hi*hi + hi*lo + lo*hi by
// THREE v_mfma_f32_16x16x32_f16 per fp32-equivalent tile step into ONE float32 accumulator (the lo*lo term, 2^-22
// relative, is dropped; the matrix pipe handles fp16 subnormals exactly - archive/proto/denorm_mfma.hip).  The f16 matrix pipe
// runs at 16x the rate of v_mfma_f32_16x16x4_f32, so the layer costs 3/16 of the exact-fp32 form.  Measured against
// float64 on random data a 256-term contraction is MORE accurate than the fp32 fma chain (1.9e-7 vs 2.9e-7 rel-L2: exact
// products, 8x fewer roundings per accumulator - archive/proto/split_mfma.hip); what it does NOT reproduce is the last bit of
// a quarter of the weights (a 13-bit residual in an 11-bit lo), a FIXED perturbation of <= 2^-23 relative that the 26
// policy evaluations of a rollout see coherently: the 25-step policy gradient sits at 4e-6 of the float64 oracle instead
// of 1.2e-6 (tools/engine_error.py; the bar is 1e-4, tests/yardstick.py states the allowance).
// Scaling (powers of two, exact): stationary weights are carried as W * 64 and activations enter the image as x * 16 so
// that typical magnitudes sit well inside fp16's normal range (absolute floor 2^-25/16 per element); gradients entering
// the reverse layer are additionally scaled per row (backward_dz2).  -DMPG_F32_MFMA selects the exact fp32 engine
// (v_mfma_f32_16x16x4_f32, k-ordered fma chain) of round 1 instead; both hold the stationary kernel in 128 registers/wave.
#ifndef MPG_F32_MFMA
#define MPG_SPLIT 1
#endif
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// LDS image of the split engine (halves): element (row, k), k = 32 kb + 8 kg + j, lives at kg*PLANE_H + row*ROW_H + 8 kb + j -
// four planes by kg (plane stride = 0 mod 64 dwords), rows 36 dwords apart (9 mod 16 in units of 4 banks): the reader's
// ds_read_b128 (lane = (row, kg), 8 consecutive k) is conflict-free in each of the instruction's four 16-lane groups (every
// group holds each row once: MI355X_MICROARCH.md §LDS), the writer's ds_write_b32 pairs are 2-way (free).  Two images: hi, lo.
constexpr int ROW_H = 72, PLANE_H = GROUP * ROW_H, IMG_H = 4 * PLANE_H;      // 4608 halves = 9216 B per image
__device__ __forceinline__ int h_index(int row, int k) { return ((k >> 3) & 3) * PLANE_H + row * ROW_H + 8 * (k >> 5) + (k & 7); }
// (A TRANSPOSED image read back through ds_read_b64_tr_b16 was built in round 4 and measured null on the sweeps: it lives in
// archive/proto/ablation_macros.patch with the other experiment branches.)
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr float W_SCALE = 64.f;          // stationary weights are stored as W * 64
constexpr float A_SCALE = 16.f;          // activations enter the LDS images as x * 16
// Envelope of the split engine (include/mpg_hip.h, "Numerical envelope"): a first-layer activation at or beyond H_LIMIT would
// enter the image as an fp16 infinity.  Every forward pass folds its first-layer pre-activations into a running maximum (four
// v_max3 per row group: for z > 0 the ELU is the identity, and it never goes below -1) and the kernel reports once, at its end.
constexpr float H_LIMIT = 65504.f / A_SCALE;      // 4094
constexpr float P_LIMIT = 65504.f / W_SCALE;      // 1023.5
__device__ __forceinline__ float max8(const float (&a)[4], const float (&b)[4], float m) {
    m = __builtin_fmaxf(__builtin_fmaxf(a[0], a[1]), m);     // the compiler fuses each pair into one v_max3_f32
    m = __builtin_fmaxf(__builtin_fmaxf(a[2], a[3]), m);
    m = __builtin_fmaxf(__builtin_fmaxf(b[0], b[1]), m);
    m = __builtin_fmaxf(__builtin_fmaxf(b[2], b[3]), m);
    return m;
}
// one lane of the wave ORs the bit: the word is sticky, the caller reads and clears it
__device__ __forceinline__ void report_activation_range(int* status, float zmax) {
    if (status && !(zmax < H_LIMIT)) atomicOr(status, MPG_STATUS_ACTIVATION_RANGE);
}

// (w0, w1) -> the two packed fp16 words: hi pair and lo pair (element 0 in the low half)
__device__ __forceinline__ void split_pack2(float w0, float w1, float& hi_word, float& lo_word) {
    const f16x2 hi = {(_Float16)w0, (_Float16)w1};
    const f16x2 lo = {(_Float16)(w0 - (float)hi[0]), (_Float16)(w1 - (float)hi[1])};
    hi_word = __builtin_bit_cast(float, hi);
    lo_word = __builtin_bit_cast(float, lo);
}

// stationary weights: W * W_SCALE, clamped to the fp16 range (a parameter beyond the envelope is reported where the images
// are packed - weight_cache.hip, optim_kernels.hip; this strided path only keeps it finite)
__device__ __forceinline__ float w_scaled(float w) { return fminf(fmaxf(w * W_SCALE, -65504.f), 65504.f); }

struct Net {
    const float *W1, *b1, *W2, *b2, *W3, *b3;
    int in_dim, out_dim;
};

__host__ __device__ inline int net_size(int in_dim, int out_dim) {
    return in_dim * H + H + H * H + H + H * out_dim + out_dim;
}

__host__ __device__ inline Net make_net(const float* p, int in_dim, int out_dim) {
    Net n;
    n.in_dim = in_dim;
    n.out_dim = out_dim;
    n.W1 = p;
    n.b1 = n.W1 + in_dim * H;
    n.W2 = n.b1 + H;
    n.b2 = n.W2 + H * H;
    n.W3 = n.b2 + H;
    n.b3 = n.W3 + H * out_dim;
    return n;
}

struct Lane {
    int lane, wave, c, rg;     // c = lane & 15 (column in tile), rg = lane >> 4 (row quad)
    __device__ Lane() {
        lane = threadIdx.x & 63;
        wave = threadIdx.x >> 6;
        c = lane & 15;
        rg = lane >> 4;
    }
    __device__ int col(int t) const { return 32 * wave + 16 * t + c; }
    __device__ int row(int j) const { return 4 * rg + j; }
};

// Static issue priority for the second-dispatched half of the workgroup.  The two waves of a SIMD (w and w + 4) share its
// matrix pipe and vector issue, arbitrated by priority, then AGE: at equal priority the younger wave loses every
// arbitration, leaves each matrix block last and is the one the step's barriers wait for.  One s_setprio for waves 4..7 at
// kernel entry (never flipped) evens the pair out: reverse sweep 81.9 -> 80.6 us, forward sweep 65.3 -> 64.8 us
// (tools/ab_scale.sh; levels 3 for the young half or 1 for the OLD half are slower; in the critic / target / network kernels
// it made no difference or a negative one and is not used).
__device__ __forceinline__ void prefer_young_waves() {
    if (threadIdx.x >= 256) __builtin_amdgcn_s_setprio(1);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL store
// (s_waitcnt vmcnt(0)), which puts the HBM write latency of the activation stash on the serial chain of every
// rollout step; nothing inside the engine kernels communicates through global memory, so LDS ordering is all that
// is needed.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef MPG_TIMELINE   // diagnostic build only (tools/timeline.sh): s_memtime of every wave at marked points of workgroup 0 / 100
#define MPG_TL_MARKS 24
__shared__ unsigned long long s_tl[8][MPG_TL_MARKS];
#define MPG_TL_DECL
#define MPG_TL(k) do { __builtin_amdgcn_sched_barrier(0); if ((threadIdx.x & 63) == 0) s_tl[threadIdx.x >> 6][k] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define MPG_TL_DUMP(dbg) do { __syncthreads(); if (dbg && (blockIdx.x == 0 || blockIdx.x == 100) && threadIdx.x < NWAVE * MPG_TL_MARKS) \
    dbg[(blockIdx.x ? 1 : 0) * NWAVE * MPG_TL_MARKS + threadIdx.x] = s_tl[threadIdx.x / MPG_TL_MARKS][threadIdx.x % MPG_TL_MARKS]; } while (0)
#else
#define MPG_TL_DECL
#define MPG_TL(k)
#define MPG_TL_DUMP(dbg)
#endif

#ifdef MPG_STAMP   // diagnostic build only: per-wave cycle accounting of the step phases (tools/stamp.sh)
__shared__ unsigned long long g_st_acc[NWAVE][10];
__shared__ unsigned long long g_st_prev[NWAVE];
__device__ __forceinline__ void mpg_stamp(int k) {
    __builtin_amdgcn_sched_barrier(0);
    if ((threadIdx.x & 63) == 0) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        const int w = threadIdx.x >> 6;
        g_st_acc[w][k] += now - g_st_prev[w];
        g_st_prev[w] = now;
    }
    __builtin_amdgcn_sched_barrier(0);
}
#define MPG_STAMP_AT(k) mpg_stamp(k)
#else
#define MPG_STAMP_AT(k)
#endif

// ELU(z) = z for z > 0, exp(z) - 1 otherwise.  exp(z) - 1 >= z everywhere, so the selection is the MEDIAN of
// (z, exp(z) - 1, 0): one v_med3_f32 instead of compare + select (same values bit for bit).
__device__ __forceinline__ float elu(float z) { return __builtin_amdgcn_fmed3f(z, __expf(z) - 1.f, 0.f); }
// ELU of the 8 values of a lane, stage by stage (all multiplies, all exps, all adds, all selects): eight independent
// chains keep the quarter-rate exp pipe busy; element by element through one temporary - what the scheduler picks at
// this register pressure - every v_exp_f32 is followed by a hazard wait
__device__ __forceinline__ void elu8(const f32x4& z0, const f32x4& z1, float (&h)[2][4]) {
    float e[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { e[j] = z0[j] * 1.4426950408889634f; e[4 + j] = z1[j] * 1.4426950408889634f; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = __builtin_amdgcn_exp2f(e[i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] -= 1.f;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[0][j] = __builtin_amdgcn_fmed3f(z0[j], e[j], 0.f);
        h[1][j] = __builtin_amdgcn_fmed3f(z1[j], e[4 + j], 0.f);
    }
}

// ELU'(z) expressed through the stored output h = ELU(z): 1 for z > 0, exp(z) = h + 1 in (0, 1] otherwise, i.e. the
// median of (h + 1, 1, 0).  (fminf(h, 0) + 1 costs an extra v_max canonicalisation of its operand.)
__device__ __forceinline__ float elu_grad_from_out(float h) { return __builtin_amdgcn_fmed3f(h + 1.f, 1.f, 0.f); }

// ---- non-finite inputs ----------------------------------------------------------------------------------------------
// Learner-side judge_is_nan (worker.py:95-107 on every worker obs / action, optimizer.py:357-361 on the gradient list).  The
// ELU's v_med3_f32 DROPS a NaN (the median of (NaN, NaN, 0) is 0), so a NaN in a replay row would leave these kernels as a
// finite, wrong value that neither the status word nor the gradient NaN guard can see.  Two things restore the reference's
// behaviour: every thread that loads a network input ORs MPG_STATUS_NAN into the caller's status word when it sees one, and
// the row's result is poisoned - row_poison() is 0 for a finite row and NaN for a row with a NaN (or an infinity: 0 * inf)
// among its n inputs; added to the row's TD error / target value it makes the gradient non-finite, which the clip kernel
// turns into a zeroed step exactly like the reference (`grads = [tf.zeros_like(grad) ...]`).
__device__ __forceinline__ float row_poison(const float* sXrow, int n) {
    float z = 0.f;
    for (int i = 0; i < n; ++i) z = fmaf(sXrow[i], 0.f, z);
    return z;
}
__device__ __forceinline__ void report_nan(int* status, bool saw_nan) {
    if (status && saw_nan) atomicOr(status, MPG_STATUS_NAN);
}

// ---- cross-lane sum over the 16 lanes of a DPP row (lanes sharing l>>4) -------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_allreduce16(float v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    v += dpp_mov<0x140>(v);   // row_mirror
    return v;
}

// ---- LDS A image --------------------------------------------------------------------------------------
// element (row, k) lives at row*LDA + (k&3)*KS + (k>>2): the MFMA A fragment of lane (row = l&15, kq = l>>4)
// for k-steps q..q+3 (k = 4q + kq) is then ONE aligned 16-byte read.
__device__ __forceinline__ int a_index(int row, int k) { return row * LDA + (k & 3) * KS + (k >> 2); }

// float32 image (exact engine; always used for the small dx product of the reverse pass)
__device__ __forceinline__ void store_c_to_a_f32(float* sA, const Lane& L, const float (&v)[2][4]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) sA[a_index(L.row(j), L.col(t))] = v[t][j];
}

#ifdef MPG_SPLIT
// Split engine: two fp16 images (hi, then lo; layout h_index) in the same LDS region.  A lane owns (rows 4rg..4rg+3, column c)
// of each tile; columns c and c^1 are adjacent k, so the pair of lanes exchanges values through one DPP quad_perm and each
// lane writes packed (k even, k odd) words: even lanes the rows j = 0,1, odd lanes the rows j = 2,3 - 8 ds_write_b32 per
// lane like the float32 image, and the reader's 8 consecutive k are one aligned 16-byte read per image.
__device__ __forceinline__ void store_c_to_a(float* sA, const Lane& L, const float (&v)[2][4]) {
    _Float16* sH = reinterpret_cast<_Float16*>(sA);
    const bool odd = L.c & 1;
    const int row0 = 4 * L.rg + (odd ? 2 : 0);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float p[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) p[j] = dpp_mov<0xB1>(v[t][j]);               // partner lane (c ^ 1)
        const int k = 32 * L.wave + 16 * t + (L.c & ~1);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float x = odd ? p[2 + u] : v[t][u], y = odd ? v[t][2 + u] : p[u];   // (k even, k odd) of row row0 + u
            float hi, lo;
            split_pack2(x * A_SCALE, y * A_SCALE, hi, lo);
            *reinterpret_cast<float*>(sH + h_index(row0 + u, k)) = hi;
            *reinterpret_cast<float*>(sH + IMG_H + h_index(row0 + u, k)) = lo;
        }
    }
}
#else
__device__ __forceinline__ void store_c_to_a(float* sA, const Lane& L, const float (&v)[2][4]) { store_c_to_a_f32(sA, L, v); }
#endif

// ---- stationary hidden kernel -------------------------------------------------------------------------
#ifdef MPG_SPLIT
// Register image of the split engine: w[4v + r], v = (kb*2 + t)*2 + part (part 0: hi, 1: lo), r = 0..3, holds the packed
// pair (k0, k0 + 1), k0 = 32 kb + 8 (l>>4) + 2 r, of output column n = 32 w + 16 t + (l&15): w[4v .. 4v+3] is the B
// fragment (8 consecutive k) of v_mfma_f32_16x16x32_f16.  Stored value: W * W_SCALE, split.
// forward:  B[k][n] = W2[k][n]
__device__ __forceinline__ void load_w2_fwd(const float* __restrict__ W2, const Lane& L, float (&w)[128]) {
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k0 = 32 * kb + 8 * L.rg + 2 * r;
                split_pack2(w_scaled(W2[k0 * H + L.col(t)]), w_scaled(W2[(k0 + 1) * H + L.col(t)]),
                            w[4 * ((kb * 2 + t) * 2) + r], w[4 * ((kb * 2 + t) * 2 + 1) + r]);
                if (r == 3) __builtin_amdgcn_sched_barrier(0);      // 8 loads in flight, not 256: this is the slow (uncached) path
            }
}
// backward: dh1[row][k1] = sum_n dz2[row][n] W2[k1][n]  ->  B[contraction n][output k1] = W2[k1][n]
__device__ __forceinline__ void load_w2_bwd(const float* __restrict__ W2, const Lane& L, float (&w)[128]) {
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k0 = 32 * kb + 8 * L.rg + 2 * r;
                split_pack2(w_scaled(W2[L.col(t) * H + k0]), w_scaled(W2[L.col(t) * H + k0 + 1]),
                            w[4 * ((kb * 2 + t) * 2) + r], w[4 * ((kb * 2 + t) * 2 + 1) + r]);
                if (r == 3) __builtin_amdgcn_sched_barrier(0);
            }
}
// Position (in 1 KiB blocks of 64 lanes x 16 bytes) of wave `wave`'s v-th block in a packed image: one contiguous 32 KB run per wave.
// Every workgroup of a launch loads the same 256 KB image at the same time (~9 k cycles, ~45 % of what the L2 could deliver).  Round 4
// tried the eight waves' blocks INTERLEAVED (v * 8 + wave: the eight streams of a CU then read consecutive blocks instead of sitting
// 32 KB apart) - it is SLOWER: bench step 0.2305 -> 0.2423 ms, the worker launch 21.4 -> 30.4 us; runs of 2 / 4 KB and a per-wave phase
// shift: within the noise; runs of 8 KB: forward sweep -0.9 us three times out of three, the step within the noise (tools/ab_img.sh).
// (the other layouts: archive/proto/ablation_macros.patch)
__host__ __device__ constexpr int img_slot(int wave, int v) {
    return wave * 32 + v;
}
__host__ __device__ inline void img_unslot(int slot, int& wave, int& v) {
    wave = slot >> 5; v = slot & 31;
}
// Same register images from the pre-packed copy kept by the weight cache (weight_cache.hip): 16-byte word index
// img_slot(wave, v)*64 + lane holds w[4v .. 4v+3] -> 32 fully coalesced 1 KiB loads per wave.
__device__ __forceinline__ void load_w2_packed(const float* __restrict__ pack, const Lane& L, float (&w)[128]) {
    const f32x4* p = reinterpret_cast<const f32x4*>(pack) + L.lane;
#pragma unroll
    for (int v = 0; v < 32; ++v) {
        const f32x4 q = p[img_slot(L.wave, v) * 64];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[4 * v + e] = q[e];
    }
}
#else
// forward:  B[k][n] = W2[k][n];  lane holds, for k-step q and tile t, W2[4q + (l>>4)][32w + 16t + (l&15)]
__device__ __forceinline__ void load_w2_fwd(const float* __restrict__ W2, const Lane& L, float (&w)[128]) {
#pragma unroll
    for (int q = 0; q < 64; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t) w[2 * q + t] = W2[(4 * q + L.rg) * H + L.col(t)];
}
// backward: dh1[row][k] = sum_n dz2[row][n] W2[k][n]  ->  B[n][k] = W2[k][n]
__device__ __forceinline__ void load_w2_bwd(const float* __restrict__ W2, const Lane& L, float (&w)[128]) {
#pragma unroll
    for (int q = 0; q < 64; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t) w[2 * q + t] = W2[L.col(t) * H + 4 * q + L.rg];
}

// Same register images from the pre-packed copy kept by the weight cache (weight_cache.hip): float4 index
// ((wave*16 + q/4)*2 + t)*64 + lane, entry q%4 -> 32 fully coalesced 1 KiB loads per wave instead of 128 strided ones.
__device__ __forceinline__ void load_w2_packed(const float* __restrict__ pack, const Lane& L, float (&w)[128]) {
    const f32x4* p = reinterpret_cast<const f32x4*>(pack) + (L.wave * 32) * 64 + L.lane;
#pragma unroll
    for (int q4 = 0; q4 < 16; ++q4)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 v = p[(q4 * 2 + t) * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[2 * (4 * q4 + e) + t] = v[e];
        }
}
#endif

// host side: packed copy of W2 for direction dir (0 forward, 1 backward) if `W2` is the hidden kernel of a network
// covered by one of the caller's weight-cache descriptors (cfg->wcache[], include/mpg_hip.h), else nullptr.  Pure
// functions of their arguments: the library keeps no binding table.
const float* weight_cache_lookup(const mpg_cfg_t* cfg, const float* W2, int dir);
const float* wcache_lookup(const mpg_wcache_t* wc, const float* W2, int dir);
int wcache_w2_offset(const mpg_wcache_t* wc, int k);

#ifdef MPG_SPLIT
// 16 x 256 (LDS A images hi / lo) times the wave's stationary 256 x 32 slice: 8 k-blocks x 2 tiles x 3 f16 MFMAs.
// acc0 / acc1 come in holding what is to be ADDED to the product (bias or zero) and leave holding the result.
__device__ __forceinline__ void mfma_16x256x32(const float* sA, const Lane& L, const float (&w)[128], f32x4& acc0,
                                               f32x4& acc1) {
    const _Float16* bh = reinterpret_cast<const _Float16*>(sA) + L.rg * PLANE_H + L.c * ROW_H;   // hi image: row l&15, k = 8 (l>>4) + ..
    const _Float16* bl = bh + IMG_H;                                                             // lo image
    constexpr int NKB = 8;
    auto frag = [&](int v) {
        return __builtin_bit_cast(f16x8, f32x4{w[4 * v], w[4 * v + 1], w[4 * v + 2], w[4 * v + 3]});
    };
    f32x4 m0 = {0.f, 0.f, 0.f, 0.f}, m1 = m0;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const f16x8 ah = *reinterpret_cast<const f16x8*>(bh + 8 * kb), al = *reinterpret_cast<const f16x8*>(bl + 8 * kb);
        const int v0 = (kb * 2 + 0) * 2, v1 = (kb * 2 + 1) * 2;
        m0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, frag(v0), m0, 0, 0, 0);
        m1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, frag(v1), m1, 0, 0, 0);
        m0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, frag(v0 + 1), m0, 0, 0, 0);
        m1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, frag(v1 + 1), m1, 0, 0, 0);
        m0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, frag(v0), m0, 0, 0, 0);
        m1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, frag(v1), m1, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        acc0[j] = fmaf(m0[j], 1.f / (W_SCALE * A_SCALE), acc0[j]);
        acc1[j] = fmaf(m1[j], 1.f / (W_SCALE * A_SCALE), acc1[j]);
    }
}
#else
// 16 x 256 (LDS A image) times the wave's stationary 256 x 32 slice; 128 MFMAs, two independent accumulators.
__device__ __forceinline__ void mfma_16x256x32(const float* sA, const Lane& L, const float (&w)[128], f32x4& acc0,
                                               f32x4& acc1) {
    const float* base = sA + L.c * LDA + L.rg * KS;
    constexpr int NQ4 = 16;
    // explicit one-block-ahead prefetch of the A fragments: the LDS latency of block q4+1 hides under the 8 MFMAs
    // (256 cycles) of block q4 instead of stalling the first MFMA of every block
    f32x4 a = *reinterpret_cast<const f32x4*>(base);
#pragma unroll
    for (int q4 = 0; q4 < NQ4; ++q4) {
        f32x4 nxt = a;
        if (q4 + 1 < NQ4) nxt = *reinterpret_cast<const f32x4*>(base + 4 * (q4 + 1));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], w[2 * (4 * q4 + i)], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], w[2 * (4 * q4 + i) + 1], acc1, 0, 0, 0);
        }
        a = nxt;
    }
}
#endif

// ---- small per-lane stationary pieces -----------------------------------------------------------------
template <int IN, int OU>
struct SmallRegs {
    float w1p[l1_steps<IN>()][2];   // layer-1 MFMA B operand: W1[4q + rg][col(t)] (0 beyond in_dim), index [q][t]
    float w1t[8];      // dx MFMA B operand: W1[c][32w + 4q + rg] (0 for c >= in_dim), q = 0..7
    float w1t2[IN > 16 ? 8 : 1];   // the same for input columns 16 + c (24-wide networks)
    float b1[2], b2[2];
    float w3[2][OU];   // W3[col(t)][o]
};

template <int IN, int OU>
__device__ __forceinline__ void load_small(const Net& n, const Lane& L, SmallRegs<IN, OU>& r) {
    static_assert(IN <= 24, "layer-1 MFMA covers K <= 24");
    // masks by the network's actual width (== IN for the exact instantiations, <= 16 / <= 24 for the wide ones)
#pragma unroll
    for (int q = 0; q < 8; ++q) r.w1t[q] = L.c < n.in_dim ? n.W1[L.c * H + 32 * L.wave + 4 * q + L.rg] : 0.f;
    if constexpr (IN > 16) {
#pragma unroll
        for (int q = 0; q < 8; ++q) r.w1t2[q] = 16 + L.c < n.in_dim ? n.W1[(16 + L.c) * H + 32 * L.wave + 4 * q + L.rg] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int col = L.col(t);
#pragma unroll
        for (int q = 0; q < l1_steps<IN>(); ++q) r.w1p[q][t] = (4 * q + L.rg) < n.in_dim ? n.W1[(4 * q + L.rg) * H + col] : 0.f;
        r.b1[t] = n.b1[col];
        r.b2[t] = n.b2[col];
#pragma unroll
        for (int o = 0; o < OU; ++o) r.w3[t][o] = n.W3[col * n.out_dim + o];
    }
}

// dx partial of this wave's 32 hidden columns on the matrix pipe (exact fp32 MFMA): A = dz1 (a0 | a1: this wave's own columns of the
// float32 LDS image), B = W1^T; one 16-column tile of input columns, a second one for the 24-wide networks
template <int IN, int OU>
__device__ __forceinline__ void dx_partials(const SmallRegs<IN, OU>& r, const Lane& L, const f32x4& a0, const f32x4& a1, float* sPartX) {
    f32x4 dx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) dx = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q], r.w1t[q], dx, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) dx = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q], r.w1t[4 + q], dx, 0, 0, 0);
    if (L.c < xs_of<IN>()) {        // columns beyond in_dim carry zero weights: they are written as zeros
#pragma unroll
        for (int j = 0; j < 4; ++j) sPartX[(L.wave * GROUP + L.row(j)) * xs_of<IN>() + L.c] = dx[j];
    }
    if constexpr (IN > 16) {
        f32x4 dx2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) dx2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[q], r.w1t2[q], dx2, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) dx2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[q], r.w1t2[4 + q], dx2, 0, 0, 0);
        if (16 + L.c < xs_of<IN>()) {
#pragma unroll
            for (int j = 0; j < 4; ++j) sPartX[(L.wave * GROUP + L.row(j)) * xs_of<IN>() + 16 + L.c] = dx2[j];
        }
    }
}

// ---- G16 stash ------------------------------------------------------------------------------------------
__device__ __forceinline__ void stash_store(float* __restrict__ base, long group, const Lane& L, const float (&v)[2][4]) {
    f32x4* p = reinterpret_cast<f32x4*>(base) + (group * 16 + 2 * L.wave) * 64 + L.lane;
    p[0] = f32x4{v[0][0], v[0][1], v[0][2], v[0][3]};
    p[64] = f32x4{v[1][0], v[1][1], v[1][2], v[1][3]};
}
__device__ __forceinline__ void stash_load(const float* __restrict__ base, long group, const Lane& L, float (&v)[2][4]) {
    const f32x4* p = reinterpret_cast<const f32x4*>(base) + (group * 16 + 2 * L.wave) * 64 + L.lane;
    const f32x4 a = p[0], b = p[64];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[0][j] = a[j];
        v[1][j] = b[j];
    }
}

// ---- forward through the two hidden layers + partial output layer for one row group ---------------------
// sX  [16][XS]  inputs (already scaled), sA the LDS A image, sPart [NWAVE][16][MAXOUT] output partials.
// On return h1/h2 hold this lane's C-layout activations and sPart the per-wave partial sums of h2*W3 (no bias);
// the caller must have synchronised sX before the call and may read sPart right after (ends on a barrier).
template <int IN, int OU, bool FINAL_BARRIER = true>
__device__ __forceinline__ void forward_group(const float* sX, float* sA, float* sPart, const Lane& L,
                                              const float (&w2)[128], const SmallRegs<IN, OU>& r,
                                              float (&h1)[2][4], float (&h2)[2][4], float* h1_stash = nullptr,
                                              long stash_group = 0, const float* xa_regs = nullptr, float* zmax = nullptr) {
    {   // layer 1 on the matrix pipe too: K = in_dim zero-padded to 8 (2 k-steps) or 16 (4) - sX rows are zero beyond in_dim.
        // xa_regs (optional): this lane's A operand x[row l&15][4q + (l>>4)] already in registers (no LDS round trip)
        f32x4 z0 = {r.b1[0], r.b1[0], r.b1[0], r.b1[0]}, z1 = {r.b1[1], r.b1[1], r.b1[1], r.b1[1]};
#pragma unroll
        for (int q = 0; q < l1_steps<IN>(); ++q) {
            const float xa = xa_regs ? xa_regs[q] : sX[L.c * xs_of<IN>() + 4 * q + L.rg];
            z0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, r.w1p[q][0], z0, 0, 0, 0);
            z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, r.w1p[q][1], z1, 0, 0, 0);
        }
        elu8(z0, z1, h1);
        if (zmax) *zmax = max8(h1[0], h1[1], *zmax);
    }
    store_c_to_a(sA, L, h1);
    // h1 is final here: its stash goes out now and drains under the MFMA block instead of queueing behind the h2
    // stash of all eight waves at the end of the step (the CU's store path moves 64 B/clk)
    if (h1_stash) stash_store(h1_stash, stash_group, L, h1);
    MPG_STAMP_AT(1);
    lds_barrier();
    MPG_STAMP_AT(2);
    f32x4 acc0 = {r.b2[0], r.b2[0], r.b2[0], r.b2[0]};
    f32x4 acc1 = {r.b2[1], r.b2[1], r.b2[1], r.b2[1]};
    mfma_16x256x32(sA, L, w2, acc0, acc1);
    MPG_STAMP_AT(3);
    elu8(acc0, acc1, h2);
    {   // output-layer partials: all 4*OU row sums advance stage by stage so that the DPP latencies interleave
        float p[OU][4];
#pragma unroll
        for (int o = 0; o < OU; ++o)
#pragma unroll
            for (int j = 0; j < 4; ++j) p[o][j] = fmaf(h2[1][j], r.w3[1][o], h2[0][j] * r.w3[0][o]);
        // one v_add_f32_dpp per value and stage (the compiler's own lowering is v_mov_b32_dpp + a packed add: 1.5
        // instructions per value).  volatile keeps the stage-major order, which also keeps every DPP read >= 8
        // instructions behind the write of its operand (the 2-wait-state VALU->DPP hazard is not checked inside asm).
#define MPG_DPP_STAGE(MODS)                                     \
        _Pragma("unroll") for (int o = 0; o < OU; ++o)           \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)            \
            asm volatile("v_add_f32_dpp %0, %1, %1 " MODS " row_mask:0xf bank_mask:0xf" : "=v"(p[o][j]) : "v"(p[o][j]));
        // the first stage's operands come from ordinary fmas the scheduler could place right in front of a DPP read:
        // one wait that depends on ALL of them (so every producer is ahead of it) covers the whole stage
        if constexpr (OU == 2)
            asm volatile("s_nop 1" : "+v"(p[0][0]), "+v"(p[0][1]), "+v"(p[0][2]), "+v"(p[0][3]), "+v"(p[OU - 1][0]),
                         "+v"(p[OU - 1][1]), "+v"(p[OU - 1][2]), "+v"(p[OU - 1][3]));
        else
            asm volatile("s_nop 1" : "+v"(p[0][0]), "+v"(p[0][1]), "+v"(p[0][2]), "+v"(p[0][3]));
        MPG_DPP_STAGE("quad_perm:[1,0,3,2]") MPG_DPP_STAGE("quad_perm:[2,3,0,1]")
        MPG_DPP_STAGE("row_half_mirror") MPG_DPP_STAGE("row_mirror")
#undef MPG_DPP_STAGE
        if (L.c == 0) {
#pragma unroll
            for (int o = 0; o < OU; ++o)
#pragma unroll
                for (int j = 0; j < 4; ++j) sPart[(L.wave * GROUP + L.row(j)) * MAXOUT + o] = p[o][j];
        }
    }
    MPG_STAMP_AT(4);
    if (FINAL_BARRIER) {      // callers that have work to hide behind the slower waves issue the barrier themselves
        lds_barrier();
        MPG_STAMP_AT(5);
    }
}

// Two row groups through ONE pair of barriers: layer 1 of both, barrier, the matrix block of both (the same register image,
// back to back on the matrix pipe), both epilogues, barrier.  A kernel that applies one image to several groups spends most
// of a group's ~4.9k cycles in barrier waits and LDS round trips, not in the 768 cycles of its matrix block; paired, the
// second group rides in the first one's latencies.  Same arithmetic per group as forward_group (bit-identical results).
template <int IN, int OU>
__device__ __forceinline__ void forward_group2(const float* sXa, const float* sXb, float* sAa, float* sAb, float* sPartA,
                                               float* sPartB, const Lane& L, const float (&w2)[128], const SmallRegs<IN, OU>& r,
                                               float (&h1a)[2][4], float (&h2a)[2][4], float (&h1b)[2][4], float (&h2b)[2][4],
                                               float* zmax = nullptr) {
    auto layer1 = [&](const float* sX, float* sA, float (&h1)[2][4]) {
        f32x4 z0 = {r.b1[0], r.b1[0], r.b1[0], r.b1[0]}, z1 = {r.b1[1], r.b1[1], r.b1[1], r.b1[1]};
#pragma unroll
        for (int q = 0; q < l1_steps<IN>(); ++q) {
            const float xa = sX[L.c * xs_of<IN>() + 4 * q + L.rg];
            z0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, r.w1p[q][0], z0, 0, 0, 0);
            z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, r.w1p[q][1], z1, 0, 0, 0);
        }
        elu8(z0, z1, h1);
        if (zmax) *zmax = max8(h1[0], h1[1], *zmax);
        store_c_to_a(sA, L, h1);
    };
    auto output = [&](float* sPart, const float (&h2)[2][4]) {
        float p[OU][4];
#pragma unroll
        for (int o = 0; o < OU; ++o)
#pragma unroll
            for (int j = 0; j < 4; ++j) p[o][j] = fmaf(h2[1][j], r.w3[1][o], h2[0][j] * r.w3[0][o]);
#define MPG_DPP_STAGE(MODS)                                     \
        _Pragma("unroll") for (int o = 0; o < OU; ++o)           \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)            \
            asm volatile("v_add_f32_dpp %0, %1, %1 " MODS " row_mask:0xf bank_mask:0xf" : "=v"(p[o][j]) : "v"(p[o][j]));
        if constexpr (OU == 2)
            asm volatile("s_nop 1" : "+v"(p[0][0]), "+v"(p[0][1]), "+v"(p[0][2]), "+v"(p[0][3]), "+v"(p[OU - 1][0]),
                         "+v"(p[OU - 1][1]), "+v"(p[OU - 1][2]), "+v"(p[OU - 1][3]));
        else
            asm volatile("s_nop 1" : "+v"(p[0][0]), "+v"(p[0][1]), "+v"(p[0][2]), "+v"(p[0][3]));
        MPG_DPP_STAGE("quad_perm:[1,0,3,2]") MPG_DPP_STAGE("quad_perm:[2,3,0,1]")
        MPG_DPP_STAGE("row_half_mirror") MPG_DPP_STAGE("row_mirror")
#undef MPG_DPP_STAGE
        if (L.c == 0) {
#pragma unroll
            for (int o = 0; o < OU; ++o)
#pragma unroll
                for (int j = 0; j < 4; ++j) sPart[(L.wave * GROUP + L.row(j)) * MAXOUT + o] = p[o][j];
        }
    };
    layer1(sXa, sAa, h1a);
    layer1(sXb, sAb, h1b);
    lds_barrier();
    f32x4 a0 = {r.b2[0], r.b2[0], r.b2[0], r.b2[0]}, a1 = {r.b2[1], r.b2[1], r.b2[1], r.b2[1]};
    f32x4 b0 = a0, b1 = a1;
    mfma_16x256x32(sAa, L, w2, a0, a1);
    mfma_16x256x32(sAb, L, w2, b0, b1);
    elu8(a0, a1, h2a);
    elu8(b0, b1, h2b);
    output(sPartA, h2a);
    output(sPartB, h2b);
    lds_barrier();
}

// sum of the 8 per-wave partials + bias for (row, o)
__device__ __forceinline__ float out_preact(const float* sPart, float bias, int row, int o) {
    float z = bias;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) z += sPart[(w * GROUP + row) * MAXOUT + o];
    return z;
}
// same sum as a depth-3 tree (the rollout's serial chain: 3 dependent adds instead of 8)
__device__ __forceinline__ float out_preact_tree(const float* sPart, float bias, int row, int o) {
    float p[NWAVE];
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) p[w] = sPart[(w * GROUP + row) * MAXOUT + o];
    return (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) + bias;
}

// sD3 [MAXOUT][16]: dL/dz3 of the used outputs, OUTPUT-major so that the four rows a lane needs (4 rg .. 4 rg + 3) are
// one aligned 16-byte read per output and pair up for packed fmas without register shuffles
__device__ __forceinline__ int d3_index(int row, int o) { return o * GROUP + row; }

// Split engine, reverse layer: dz2 rows span many orders of magnitude (dL/dz3 carries 1/B and the slice weights), fp16
// does not - each row of the image is scaled by 2^-e, e = exponent of max_o |dL/dz3[row][o]| (|dz2| <= ~OU |W3| |dz3|), and
// the product is scaled back by 2^e.  Powers of two: exact.  Both halves of the pass derive e from sD3 themselves.
template <int OU>
__device__ __forceinline__ void row_exponents(const float* sD3, const Lane& L, int (&e)[4]) {
    f32x4 m = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < OU; ++o) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(sD3 + o * GROUP + 4 * L.rg);     // d3_index(4 rg, o)
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], fabsf(d[j]));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_frexp_expf(m[j]);
}

// ---- backward through the hidden layers for one row group ---------------------------------------------
// sD3 [16][MAXOUT] holds dL/dz3 (pre-activation of the used outputs).  h1/h2: this lane's stashed activations.
// Produces dz2 and dz1 (C layout).  If WANT_DX, leaves per-wave partial sums of dz1*W1^T in sPartX
// [NWAVE][16][XS] and ends on a barrier; the caller reduces them.
// first half: dz2 = (dz3 W3^T) * ELU'(h2) into the LDS A image (ends on the barrier that publishes it)
template <int IN, int OU>
__device__ __forceinline__ void backward_dz2(const float* sD3, float* sA, const Lane& L, const SmallRegs<IN, OU>& r,
                                             const float (&h2)[2][4], float (&dz2)[2][4]) {
    f32x4 d3v[OU];
#pragma unroll
    for (int o = 0; o < OU; ++o) d3v[o] = *reinterpret_cast<const f32x4*>(sD3 + d3_index(4 * L.rg, o));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float dh = 0.f;
#pragma unroll
            for (int o = 0; o < OU; ++o) dh = fmaf(d3v[o][j], r.w3[t][o], dh);
            dz2[t][j] = dh * elu_grad_from_out(h2[t][j]);
        }
    }
#ifdef MPG_SPLIT
    {
        int e[4];
        row_exponents<OU>(sD3, L, e);
        float sc[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) sc[t][j] = ldexpf(dz2[t][j], -e[j]);
        store_c_to_a(sA, L, sc);
    }
#else
    store_c_to_a(sA, L, dz2);
#endif
    MPG_STAMP_AT(1);
    lds_barrier();
    MPG_STAMP_AT(2);
}

// second half: dh1 = dz2 W2^T on the matrix pipe, dz1 = dh1 * ELU'(h1), optional dx partials.  Global loads whose
// results are needed after the MFMA block (h1) or in a later step should be issued between the two halves: any
// s_waitcnt vmcnt(0) the compiler places earlier would otherwise also wait for them (the counter retires in order).
// sA1: a second LDS image (GROUP*LDA floats) for dz1 - every wave reads back only the columns it wrote itself, so no
// barrier is needed around it (re-using sA would need one: other waves may still be reading dz2 from it).
template <int IN, int OU, bool WANT_DX, bool FINAL_BARRIER = true>
__device__ __forceinline__ void backward_rest(const float* sD3, float* sA, float* sA1, float* sPartX, const Lane& L,
                                              const float (&w2t)[128], const SmallRegs<IN, OU>& r, const float (&h1)[2][4],
                                              float (&dz1)[2][4]) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    mfma_16x256x32(sA, L, w2t, acc0, acc1);
    MPG_STAMP_AT(3);
#ifdef MPG_SPLIT
    int e[4];
    row_exponents<OU>(sD3, L, e);     // sD3 is stable until the next step's barrier (not held across the MFMA block: registers)
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#ifdef MPG_SPLIT
        acc0[j] = ldexpf(acc0[j], e[j]);
        acc1[j] = ldexpf(acc1[j], e[j]);
#endif
        dz1[0][j] = acc0[j] * elu_grad_from_out(h1[0][j]);
        dz1[1][j] = acc1[j] * elu_grad_from_out(h1[1][j]);
    }
    if (WANT_DX) {
        // dx partial of this wave's 32 hidden columns on the matrix pipe (exact fp32 MFMA): A = dz1 (this wave's own
        // columns of the second LDS image, float32), B = W1^T.
        store_c_to_a_f32(sA1, L, dz1);
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): this wave reads back only what it wrote itself
        __builtin_amdgcn_wave_barrier();
        const float* base = sA1 + L.c * LDA + L.rg * KS + 8 * L.wave;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(base), a1 = *reinterpret_cast<const f32x4*>(base + 4);
        dx_partials<IN, OU>(r, L, a0, a1, sPartX);
    }
    MPG_STAMP_AT(4);
    if (FINAL_BARRIER) {
        lds_barrier();
        MPG_STAMP_AT(5);
    }
}


template <int IN, int OU, bool WANT_DX>
__device__ __forceinline__ void backward_group(const float* sD3, float* sA, float* sA1, float* sPartX, const Lane& L,
                                               const float (&w2t)[128], const SmallRegs<IN, OU>& r,
                                               const float (&h1)[2][4], const float (&h2)[2][4],
                                               float (&dz1)[2][4], float (&dz2)[2][4]) {
    backward_dz2<IN, OU>(sD3, sA, L, r, h2, dz2);
    backward_rest<IN, OU, WANT_DX>(sD3, sA, sA1, sPartX, L, w2t, r, h1, dz1);
}

// Two row groups through the reverse layer behind ONE pair of barriers (the reverse twin of forward_group2): dz2 of both into
// their images, barrier, the matrix block of both (the same transposed register image, back to back on the matrix pipe), both
// epilogues, barrier.  The float32 dz1 image of the dx product (sA1) is wave-private, so the two groups use it one after the
// other with no barrier.  Same arithmetic per group as backward_group: bit-identical results.
template <int IN, int OU, bool WANT_DX>
__device__ __forceinline__ void backward_group2(const float* sD3a, const float* sD3b, float* sAa, float* sAb, float* sA1,
                                                float* sPartXa, float* sPartXb, const Lane& L, const float (&w2t)[128],
                                                const SmallRegs<IN, OU>& r, const float (&h1a)[2][4], const float (&h2a)[2][4],
                                                const float (&h1b)[2][4], const float (&h2b)[2][4], float (&dz1a)[2][4],
                                                float (&dz2a)[2][4], float (&dz1b)[2][4], float (&dz2b)[2][4]) {
    auto dz2_phase = [&](const float* sD3, float* sA, const float (&h2)[2][4], float (&dz2)[2][4]) {
        f32x4 d3v[OU];
#pragma unroll
        for (int o = 0; o < OU; ++o) d3v[o] = *reinterpret_cast<const f32x4*>(sD3 + d3_index(4 * L.rg, o));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float dh = 0.f;
#pragma unroll
                for (int o = 0; o < OU; ++o) dh = fmaf(d3v[o][j], r.w3[t][o], dh);
                dz2[t][j] = dh * elu_grad_from_out(h2[t][j]);
            }
        }
#ifdef MPG_SPLIT
        int e[4];
        row_exponents<OU>(sD3, L, e);
        float sc[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) sc[t][j] = ldexpf(dz2[t][j], -e[j]);
        store_c_to_a(sA, L, sc);
#else
        store_c_to_a(sA, L, dz2);
#endif
    };
    dz2_phase(sD3a, sAa, h2a, dz2a);
    dz2_phase(sD3b, sAb, h2b, dz2b);
    lds_barrier();
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, b0 = a0, b1 = a0;
    mfma_16x256x32(sAa, L, w2t, a0, a1);
    mfma_16x256x32(sAb, L, w2t, b0, b1);
    auto rest = [&](const float* sD3, float* sPartX, f32x4& acc0, f32x4& acc1, const float (&h1)[2][4], float (&dz1)[2][4]) {
#ifdef MPG_SPLIT
        int e[4];
        row_exponents<OU>(sD3, L, e);
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#ifdef MPG_SPLIT
            acc0[j] = ldexpf(acc0[j], e[j]);
            acc1[j] = ldexpf(acc1[j], e[j]);
#endif
            dz1[0][j] = acc0[j] * elu_grad_from_out(h1[0][j]);
            dz1[1][j] = acc1[j] * elu_grad_from_out(h1[1][j]);
        }
        if (WANT_DX) {
            store_c_to_a_f32(sA1, L, dz1);
            __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): this wave reads back only what it wrote itself
            __builtin_amdgcn_wave_barrier();
            const float* base = sA1 + L.c * LDA + L.rg * KS + 8 * L.wave;
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(base), q1 = *reinterpret_cast<const f32x4*>(base + 4);
            dx_partials<IN, OU>(r, L, q0, q1, sPartX);
            __builtin_amdgcn_s_waitcnt(0xC07F);          // the second group's dz1 overwrites the image this wave has just read
            __builtin_amdgcn_wave_barrier();
        }
    };
    rest(sD3a, sPartXa, a0, a1, h1a, dz1a);
    rest(sD3b, sPartXb, b0, b1, h1b, dz1b);
    lds_barrier();
}

// all XS partial sums of one row: ds_read_b128 pairs in two batches of four waves (32 VGPRs in flight; all sixteen
// reads at once cost 64 VGPRs at a point where the reverse sweep has none to spare).
// NEED: how many of the XSW entries the caller uses (the observation width).  Every read fetches exactly the entries that are
// used - b128 / b64 / none for the upper four of a block of eight: with a b128 whose upper half is dead the register allocator
// overlaps that dead half with the next read's destination, and the write-after-write hazard puts an `s_waitcnt lgkmcnt(0)`
// between the reads - eight serialised LDS round trips on the reverse sweep's serial chain (found in the ISA, round 5).
template <int XSW = XS, int NEED = XSW>
__device__ __forceinline__ void dx_reduce_row(const float* sPartX, int row, float (&out)[XSW]) {
#pragma unroll
    for (int i = 0; i < XSW; ++i) out[i] = 0.f;
#pragma unroll
    for (int half = 0; half < XSW / 8; ++half) {
        constexpr int NL_ALL = NEED >= XSW ? 8 : NEED;          // (only the one-block form, XSW == 8, is ever asked for fewer)
        const int nl = XSW == 8 ? NL_ALL : 8;                   // entries used in this block of eight
#pragma unroll
        for (int b = 0; b < NWAVE; b += 4) {
            f32x4 lo[4], hi[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float* p = sPartX + ((b + w) * GROUP + row) * XSW + 8 * half;
                lo[w] = *reinterpret_cast<const f32x4*>(p);
                if (nl > 6) hi[w] = *reinterpret_cast<const f32x4*>(p + 4);
                else if (nl > 4) {
                    // (as ONE 64-bit integer, taken apart by shifts: read as a two-float vector, the sums below are re-formed into
                    // v_pk_add_f32 by the vector combiner - packed fp32 arithmetic beside matrix instructions is what the containment
                    // rule of DESIGN.md section 4.2 / tests/test_abi.py keeps out of the shipped ISA)
                    const unsigned long long u = *reinterpret_cast<const unsigned long long*>(p + 4);
                    hi[w] = f32x4{__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32)), 0.f, 0.f};
                } else hi[w] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                out[8 * half + i] += (lo[0][i] + lo[1][i]) + (lo[2][i] + lo[3][i]);
                if (4 + i < nl) out[8 * half + 4 + i] += (hi[0][i] + hi[1][i]) + (hi[2][i] + hi[3][i]);
            }
        }
    }
}

template <int XSW = XS>
__device__ __forceinline__ float dx_reduce(const float* sPartX, int row, int i) {
    float z = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) z += sPartX[(w * GROUP + row) * XSW + i];
    return z;
}

}  // namespace mlp
