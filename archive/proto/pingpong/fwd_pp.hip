// Prototype (round 4): PING-PONG geometry for the chain-free forward pass.
//   workgroup = 512 threads = 8 waves = two TEAMS of four waves (team = wave >> 2: the two waves of a SIMD are w and w + 4, so
//   every SIMD hosts one wave of each team).  A team evaluates row groups on its own: wave q of a team owns 64 hidden columns
//   (4 tiles of 16); the hi halves of its 256 x 64 slice of W2 are register-stationary (128 VGPRs), the lo halves are read from a
//   112 KB LDS image shared by both teams (k-blocks 0..6; k-block 7 stays in 16 VGPRs: the whole lo image does not fit beside
//   the activation images).  The teams run half a period apart: while one issues its matrix block (96 MFMAs per wave, 1536
//   cycles of its SIMD's matrix pipe), the other does the vector work around it (epilogue of its previous group, layer 1 + ELU
//   + fp16 split + image store of its next) - ONE workgroup barrier per interval, matrix pipe and vector issue of every SIMD
//   busy at the same time with DIFFERENT waves.
// Question: does a row group cost less than the 4.3 - 4.5 k cycles it costs in every lock-step structure tried in round 3?
// Build + run:  bash archive/proto/pingpong/run.sh   (on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstring>
#include "mlp_core.h"

using namespace mlp;

// ---- the shipped geometry (k_forward's structure), as the reference --------------------------------------------------------
template <int IN, int OU>
__global__ void __launch_bounds__(NTHREAD, 2) k_fwd8(const float* params, int in_dim, int out_dim, int rows, const float* x, float* y) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_IMG + GROUP * XS + NWAVE * GROUP * MAXOUT)];
    float* sA = smem;
    float* sX = sA + 2 * A_IMG;
    float* sPart = sX + 2 * GROUP * XS;
    const Lane L;
    const Net net = make_net(params, in_dim, out_dim);
    float w2[128];
    SmallRegs<IN, OU> r;
    load_small<IN, OU>(net, L, r);
    load_w2_fwd(net.W2, L, w2);
    const long ngroups = (rows + GROUP - 1) / GROUP;
    const long nunits = (ngroups + 1) / 2;
    const int tid = threadIdx.x;
    for (long u = blockIdx.x; u < nunits; u += gridDim.x) {
        if (tid < 2 * GROUP * XS) {
            const int g2 = tid / (GROUP * XS), e = tid % (GROUP * XS), row = e / XS, i = e % XS;
            const long gr = (u * 2 + g2) * GROUP + row;
            sX[tid] = (gr < rows && i < in_dim) ? x[gr * in_dim + i] : 0.f;
        }
        lds_barrier();
        float h1[2][2][4], h2[2][2][4];
        forward_group2<IN, OU>(sX, sX + GROUP * XS, sA, sA + A_IMG, sPart, sPart + NWAVE * GROUP * MAXOUT, L, w2, r, h1[0], h2[0], h1[1], h2[1]);
        if (tid < 2 * GROUP * OU) {
            const int g2 = tid / (GROUP * OU), row = (tid / OU) % GROUP, o = tid % OU;
            const long gr = (u * 2 + g2) * GROUP + row;
            if (gr < rows) y[gr * OU + o] = out_preact(sPart + g2 * NWAVE * GROUP * MAXOUT, net.b3[o], row, o);
        }
    }
}

// ---- TRANSPOSED activation image (variant 128) ----------------------------------------------------------------------------
// One 32-byte slot per contraction index k: the 16 batch rows of the group, fp16, rows contiguous - what a C-layout lane holds
// (rows 4 rg .. 4 rg + 3 of ONE column) is one aligned 8-byte chunk, so the image store is ONE ds_write_b64 per tile and image
// (no DPP exchange, no selects).  The MFMA A operand (row l & 15, 8 consecutive k) comes back through ds_read_b64_tr_b16, the
// hardware transpose read: two reads of 4 k each per operand.  Slot order inside a k-block and an XOR on the chunk position make
// both the stores (16 lanes of a row quad) and the transposed reads (32-lane halves) bank-conflict free without padding:
//   k = 32 kb + 8 g + j  ->  slot = 32 kb + 16 (g >> 1) + 8 (j >> 2) + 4 (g & 1) + (j & 3),  chunk' = chunk ^ ((slot >> 2) & 3)
// (tr_slot / tr_byte / split2_mix / TR_IMG_BYTES now live in mlp_core.h: the product kernels of mlp_pingpong.hip use them)

// ---- ping-pong geometry --------------------------------------------------------------------------------------------------------
constexpr int TW = 4;                 // waves per team
constexpr int NT = 4;                 // 16-column tiles per wave
constexpr int LO_KB = 7;              // k-blocks of the lo image kept in LDS (the last one stays in registers)
constexpr int LO_FLOATS = TW * LO_KB * NT * 256;      // 28 672 floats = 112 KB

// AB (timing only, wrong numbers): 1 one k-block of the matrix block, 2 no exps, 4 no image stores, 8 no output reduction,
// 16 the teams in PHASE (both do their vector phase, then both their matrix phase: the lock-step control), 32 no lo reads from LDS
// 64 (correct numbers): leaner vector phase - the x16 image scale folded into layer 1 / the epilogue, image halves stored with
//    16-bit LDS stores straight from the packed conversions (no DPP exchange, no selects)
template <int IN, int OU, int AB = 0>
__global__ void __launch_bounds__(NTHREAD, 2) k_fwd_pp(const float* params, int in_dim, int out_dim, int rows, const float* x, float* y, const float* pk_hi, const float* pk_lo) {
    __shared__ __attribute__((aligned(16))) float sLo[LO_FLOATS];
    __shared__ __attribute__((aligned(16))) float sAimg[2][A_IMG];
    __shared__ float sPart[2][2][TW * GROUP * MAXOUT];          // [team][parity]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, team = wave >> 2, q = wave & 3, c = lane & 15, rg = lane >> 4;
    const Net net = make_net(params, in_dim, out_dim);
    // ---- stationary pieces ----
    float whi[128];                   // whi[((kb * NT + t) * 4) + r]: packed pair (k0, k0 + 1), k0 = 32 kb + 8 rg + 2 r, column 64 q + 16 t + c
    float wlo7[16];
    {   // packed images (host side, main()): f32x4 index ((q * 8 + kb) * NT + t) * 64 + lane
        const f32x4* ph = reinterpret_cast<const f32x4*>(pk_hi) + (q * 8 * NT) * 64 + lane;
        const f32x4* pl = reinterpret_cast<const f32x4*>(pk_lo) + (q * 8 * NT) * 64 + lane;
#pragma unroll
        for (int v = 0; v < 8 * NT; ++v) {
            const f32x4 h = ph[v * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e) whi[v * 4 + e] = h[e];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4 l = pl[(7 * NT + t) * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e) wlo7[t * 4 + e] = l[e];
        }
        // the LDS part of the lo image: wave (team, q) copies half of its slice's 28 fragments
        for (int v = team; v < LO_KB * NT; v += 2)
            *reinterpret_cast<f32x4*>(sLo + (q * LO_KB * NT + v) * 256 + lane * 4) = pl[v * 64];
    }
    float w1p[2][NT], b1[NT], b2[NT], w3[NT][OU];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = 64 * q + 16 * t + c;
        constexpr float SC = (AB & 64) ? A_SCALE : 1.f;       // leaner form: layer 1 and the hidden layer produce 16 h directly
#pragma unroll
        for (int s = 0; s < 2; ++s) w1p[s][t] = (4 * s + rg) < in_dim ? net.W1[(4 * s + rg) * H + col] * SC : 0.f;
        b1[t] = net.b1[col] * SC;
        b2[t] = net.b2[col] * SC;
#pragma unroll
        for (int o = 0; o < OU; ++o) w3[t][o] = net.W3[col * out_dim + o] / SC;
    }
    const float b3v = tid < 0 ? 0.f : net.b3[(lane % OU)];
    auto hfrag = [&](int kb, int t) {
        const int v = (kb * NT + t) * 4;
        return __builtin_bit_cast(f16x8, f32x4{whi[v], whi[v + 1], whi[v + 2], whi[v + 3]});
    };
    _Float16* sH = reinterpret_cast<_Float16*>(sAimg[team]);
    const _Float16* bh = sH + rg * PLANE_H + c * ROW_H;
    const _Float16* bl = bh + IMG_H;
    const bool odd = c & 1;
    const int row0 = 4 * rg + (odd ? 2 : 0);
    const long ngroups = (rows + GROUP - 1) / GROUP;
    // team T of workgroup b takes groups 2 b + T, 2 b + T + 2 gridDim.x, ...
    const long gstep = 2L * gridDim.x;
    const long g_first = 2L * blockIdx.x + team;
    const long n_it = (ngroups + gstep - 1) / gstep;          // iterations of EVERY team (empty groups at the end do the motions)
    auto x_load = [&](long g, float (&xa)[2]) {
        const long gr = g * GROUP + c;
#pragma unroll
        for (int s = 0; s < 2; ++s) xa[s] = (g < ngroups && gr < rows && 4 * s + rg < in_dim) ? x[gr * in_dim + 4 * s + rg] : 0.f;
    };
    f32x4 acc[NT];
    float xa[2];
    x_load(g_first, xa);
    __syncthreads();                                          // the lo image is complete
    // vector phase: epilogue of group gp (if any, from acc), layer 1 of group gn (if any)
    auto vector_phase = [&](long gp, bool have_prev, long gn, bool have_next, int par) {
        if constexpr ((AB & 256) != 0) __builtin_amdgcn_s_setprio(3);     // 256: the vector-phase wave outranks its SIMD partner's MFMA stream
        if constexpr ((AB & 512) != 0) __builtin_amdgcn_s_setprio(0);     // 512: the other way round
        if (have_prev) {
            float h2[NT][4];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (AB & 64) {
                        const float a = fmaf(acc[t][j], 1.f / W_SCALE, b2[t]);          // 16 z
                        h2[t][j] = __builtin_amdgcn_fmed3f(a, fmaf(__builtin_amdgcn_exp2f(a * (1.4426950408889634f / A_SCALE)), A_SCALE, -A_SCALE), 0.f);
                    } else {
                    const float a = fmaf(acc[t][j], 1.f / (W_SCALE * A_SCALE), b2[t]);
                    h2[t][j] = (AB & 2) ? a : __builtin_amdgcn_fmed3f(a, __builtin_amdgcn_exp2f(a * 1.4426950408889634f) - 1.f, 0.f);
                    }
                }
            float p[OU][4];
#pragma unroll
            for (int o = 0; o < OU; ++o)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float s = fmaf(h2[1][j], w3[1][o], h2[0][j] * w3[0][o]) + fmaf(h2[3][j], w3[3][o], h2[2][j] * w3[2][o]);
                    p[o][j] = (AB & 8) ? s : row_allreduce16(s);
                }
            if (c == 0) {
#pragma unroll
                for (int o = 0; o < OU; ++o)
#pragma unroll
                    for (int j = 0; j < 4; ++j) sPart[team][par][(q * GROUP + 4 * rg + j) * MAXOUT + o] = p[o][j];
            }
        }
        if (have_next) {
            f32x4 z[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) z[t] = f32x4{b1[t], b1[t], b1[t], b1[t]};
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < NT; ++t) z[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], w1p[s][t], z[t], 0, 0, 0);
            if constexpr (AB & 128) {
                char* img = reinterpret_cast<char*>(sAimg[team]);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    float h[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) h[j] = __builtin_amdgcn_fmed3f(z[t][j], __builtin_amdgcn_exp2f(z[t][j] * 1.4426950408889634f) - 1.f, 0.f);
                    unsigned h01, l01, h23, l23;
                    split2_mix(h[0], h[1], A_SCALE, h01, l01);
                    split2_mix(h[2], h[3], A_SCALE, h23, l23);
                    const int byte = tr_byte(tr_slot(64 * q + 16 * t + c), rg);
                    *reinterpret_cast<u32x2*>(img + byte) = u32x2{h01, h23};
                    *reinterpret_cast<u32x2*>(img + TR_IMG_BYTES + byte) = u32x2{l01, l23};
                }
            } else if constexpr (AB & 64) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    float h[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        h[j] = __builtin_amdgcn_fmed3f(z[t][j], fmaf(__builtin_amdgcn_exp2f(z[t][j] * (1.4426950408889634f / A_SCALE)), A_SCALE, -A_SCALE), 0.f);
                    const int k = 64 * q + 16 * t + c;
                    _Float16* ph = sH + h_index(4 * rg, k);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const _Float16 hi = (_Float16)h[j];
                        const _Float16 lo = (_Float16)(h[j] - (float)hi);
                        ph[j * ROW_H] = hi;
                        ph[IMG_H + j * ROW_H] = lo;
                    }
                }
            } else
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float h1[4], pn[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    h1[j] = (AB & 2) ? z[t][j] : __builtin_amdgcn_fmed3f(z[t][j], __builtin_amdgcn_exp2f(z[t][j] * 1.4426950408889634f) - 1.f, 0.f);
#pragma unroll
                for (int j = 0; j < 4; ++j) pn[j] = dpp_mov<0xB1>(h1[j]);
                const int k = 64 * q + 16 * t + (c & ~1);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float xx = odd ? pn[2 + u] : h1[u], yy = odd ? h1[2 + u] : pn[u];
                    float hi, lo;
                    split_pack2(xx * A_SCALE, yy * A_SCALE, hi, lo);
                    if (AB & 4) asm volatile("" :: "v"(hi), "v"(lo));
                    else {
                        *reinterpret_cast<float*>(sH + h_index(row0 + u, k)) = hi;
                        *reinterpret_cast<float*>(sH + IMG_H + h_index(row0 + u, k)) = lo;
                    }
                }
            }
        }
        (void)gp; (void)gn;
    };
    auto matrix_phase = [&](long gnext) {
        if constexpr ((AB & 256) != 0) __builtin_amdgcn_s_setprio(0);
        if constexpr ((AB & 512) != 0) __builtin_amdgcn_s_setprio(3);
        x_load(gnext, xa);                                    // the next group's inputs travel under the matrix block
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < ((AB & 1) ? 1 : 8); ++kb) {
            f16x8 ah, al;
            if constexpr (AB & 128) {
                typedef __attribute__((address_space(3))) s16x4* lds_p;
                const char* img = reinterpret_cast<const char*>(sAimg[team]);
                // lane 4 q' + p' of its 16-lane group supplies block row q' (k = 32 kb + 8 rg + q' (+ 4)), chunk p'
                const int qq = (lane >> 2) & 3, pp = lane & 3;
                const int s0 = 32 * kb + 16 * (rg >> 1) + 4 * (rg & 1) + qq, s1 = s0 + 8;
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + tr_byte(s0, pp)));
                const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + tr_byte(s1, pp)));
                const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + TR_IMG_BYTES + tr_byte(s0, pp)));
                const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + TR_IMG_BYTES + tr_byte(s1, pp)));
                typedef short s16x8 __attribute__((__vector_size__(8 * sizeof(short))));
                ah = __builtin_bit_cast(f16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
                al = __builtin_bit_cast(f16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
            } else {
                ah = *reinterpret_cast<const f16x8*>(bh + 8 * kb);
                al = *reinterpret_cast<const f16x8*>(bl + 8 * kb);
            }
            f16x8 wl[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (kb < LO_KB && !(AB & 32)) wl[t] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(sLo + ((q * LO_KB + kb) * NT + t) * 256 + lane * 4));
                else wl[t] = __builtin_bit_cast(f16x8, f32x4{wlo7[t * 4], wlo7[t * 4 + 1], wlo7[t * 4 + 2], wlo7[t * 4 + 3]});
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, hfrag(kb, t), acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, hfrag(kb, t), acc[t], 0, 0, 0);
        }
    };
    auto output = [&](long g, int par) {                      // the 16 * OU output lanes of the team's first wave
        if (q == 0 && lane < GROUP * OU && g < ngroups) {
            const int row = lane / OU, o = lane % OU;
            const long gr = g * GROUP + row;
            float zz = b3v;
#pragma unroll
            for (int w = 0; w < TW; ++w) zz += sPart[team][par][(w * GROUP + row) * MAXOUT + o];
            if (gr < rows) y[gr * OU + o] = zz;
        }
    };
    // Team 0: V0 | M0 | V1 | M1 | ... ; team 1 the same one interval later (it opens with an idle interval and team 0 closes with one).
    // One workgroup barrier per interval.  Iteration i of a team: group g_first + i * gstep.
    // 1024: TEAM-LOCAL barriers (an LDS arrival counter per team, polled with s_sleep) instead of workgroup barriers: the two teams
    // are then coupled only through the matrix pipe and the LDS they share - no team ever waits for the other one's phase to end
    __shared__ unsigned sTeamCnt[2];
    if (tid < 2) sTeamCnt[tid] = 0;
    __syncthreads();
    unsigned tb_target = 0;
    auto team_barrier = [&]() {
        tb_target += TW;
        __builtin_amdgcn_s_waitcnt(0xC07F);                                 // lgkmcnt(0): this wave's LDS stores are in
        if (lane == 0) __hip_atomic_fetch_add(&sTeamCnt[team], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(&sTeamCnt[team], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < tb_target) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    const bool inphase = (AB & 16) != 0;
    constexpr bool TEAMBAR = (AB & 1024) != 0;
    if (team == 1 && !inphase && !TEAMBAR) lds_barrier();
    for (long i = 0; i <= n_it; ++i) {
        const long g = g_first + i * gstep;
        // vector interval: epilogue of iteration i - 1, output of iteration i - 2, layer 1 of iteration i
        vector_phase(g - gstep, i > 0, g, i < n_it, (int)((i + 1) & 1));
        if (i >= 2) output(g - 2 * gstep, (int)(i & 1));
        if constexpr (TEAMBAR) team_barrier(); else lds_barrier();
        if (i == n_it) break;
        matrix_phase(g + gstep);
        if constexpr (TEAMBAR) team_barrier(); else lds_barrier();
    }
    if (team == 0 && !inphase && !TEAMBAR) lds_barrier();
    // the last two outputs: iteration n_it - 1's partials were written in the last vector interval (parity n_it & 1 ... )
    if constexpr (TEAMBAR) team_barrier(); else lds_barrier();
    if (n_it >= 1) output(g_first + (n_it - 1) * gstep, (int)((n_it + 1) & 1));
}

int main() {
    const int IN = 8, OUT = 1, rows = 65536;
    const int np = net_size(IN, OUT);
    std::vector<float> hp(np), hx((size_t)rows * IN);
    srand(1);
    auto rnd = [] { return (rand() / (float)RAND_MAX) * 2.f - 1.f; };
    for (int i = 0; i < np; ++i) hp[i] = rnd() * 0.1f;
    for (auto& v : hx) v = rnd();
    float *dp, *dx, *y8, *ypp;
    hipMalloc(&dp, np * 4); hipMalloc(&dx, hx.size() * 4); hipMalloc(&y8, rows * 4); hipMalloc(&ypp, rows * 4);
    hipMemcpy(dp, hp.data(), np * 4, hipMemcpyHostToDevice);
    hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    // packed hi / lo images in the ping-pong lane order
    std::vector<float> hh(65536), hl(65536);
    {
        const float* W2 = hp.data() + IN * 256 + 256;
        for (int q = 0; q < 4; ++q) for (int kb = 0; kb < 8; ++kb) for (int t = 0; t < 4; ++t) for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 4; ++r) {
            const int c = lane & 15, rg = lane >> 4, k0 = 32 * kb + 8 * rg + 2 * r, col = 64 * q + 16 * t + c;
            _Float16 h2[2], l2[2];
            for (int e = 0; e < 2; ++e) {
                float w = W2[(k0 + e) * 256 + col] * 64.f;
                h2[e] = (_Float16)w; l2[e] = (_Float16)(w - (float)h2[e]);
            }
            const size_t idx = ((((size_t)q * 8 + kb) * 4 + t) * 64 + lane) * 4 + r;
            memcpy(&hh[idx], h2, 4); memcpy(&hl[idx], l2, 4);
        }
    }
    float *dhh, *dhl;
    hipMalloc(&dhh, 65536 * 4); hipMalloc(&dhl, 65536 * 4);
    hipMemcpy(dhh, hh.data(), 65536 * 4, hipMemcpyHostToDevice);
    hipMemcpy(dhl, hl.data(), 65536 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rr : {65536, 16384, 8192, 4096}) {
        float t8 = 0, tpp = 0;
        hipMemset(ypp, 0, rows * 4);
        for (int which = 0; which < 2; ++which) {
            for (int it = 0; it < 20; ++it) {
                if (which == 0) hipLaunchKernelGGL((k_fwd8<8, 1>), dim3(256), dim3(NTHREAD), 0, 0, dp, IN, OUT, rr, dx, y8);
                else hipLaunchKernelGGL((k_fwd_pp<8, 1>), dim3(256), dim3(NTHREAD), 0, 0, dp, IN, OUT, rr, dx, ypp, dhh, dhl);
            }
            hipEventRecord(e0);
            for (int it = 0; it < 100; ++it) {
                if (which == 0) hipLaunchKernelGGL((k_fwd8<8, 1>), dim3(256), dim3(NTHREAD), 0, 0, dp, IN, OUT, rr, dx, y8);
                else hipLaunchKernelGGL((k_fwd_pp<8, 1>), dim3(256), dim3(NTHREAD), 0, 0, dp, IN, OUT, rr, dx, ypp, dhh, dhl);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            (which == 0 ? t8 : tpp) = ms * 10.f;
        }
        std::vector<float> a(rr), b(rr);
        hipMemcpy(a.data(), y8, rr * 4, hipMemcpyDeviceToHost);
        hipMemcpy(b.data(), ypp, rr * 4, hipMemcpyDeviceToHost);
        double md = 0, mx = 0;
        for (int i = 0; i < rr; ++i) { md = fmax(md, fabs(a[i] - b[i])); mx = fmax(mx, fabs(a[i])); }
        printf("rows %6d (%4.1f groups / workgroup): shipped pairs %.1f us   ping-pong %.1f us   max |diff| %.2e (max |y| %.2f)  err=%s\n", rr,
               rr / 16 / 256.0, t8, tpp, md, mx, hipGetErrorString(hipGetLastError()));
    }
    for (int variant = 0; variant < 3; ++variant) {
        hipMemset(ypp, 0, rows * 4);
        if (variant == 0) hipLaunchKernelGGL((k_fwd_pp<8, 1, 64>), dim3(256), dim3(NTHREAD), 0, 0, dp, IN, OUT, rows, dx, ypp, dhh, dhl);
        else if (variant == 1) hipLaunchKernelGGL((k_fwd_pp<8, 1, 128>), dim3(256), dim3(NTHREAD), 0, 0, dp, IN, OUT, rows, dx, ypp, dhh, dhl);
        else hipLaunchKernelGGL((k_fwd_pp<8, 1, 128 + 1024>), dim3(256), dim3(NTHREAD), 0, 0, dp, IN, OUT, rows, dx, ypp, dhh, dhl);
        std::vector<float> a(rows), b(rows);
        hipMemcpy(a.data(), y8, rows * 4, hipMemcpyDeviceToHost);
        hipMemcpy(b.data(), ypp, rows * 4, hipMemcpyDeviceToHost);
        double md = 0;
        for (int i = 0; i < rows; ++i) md = fmax(md, fabs(a[i] - b[i]));
        printf("%s variant vs shipped: max |diff| %.2e  err=%s\n", variant == 2 ? "team-barrier" : variant ? "transposed-image" : "lean", md, hipGetErrorString(hipGetLastError()));
    }
    auto timepp = [&](auto kern, const char* name) {
        for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(kern, dim3(256), dim3(NTHREAD), 0, 0, dp, IN, OUT, rows, dx, ypp, dhh, dhl);
        hipEventRecord(e0);
        for (int it = 0; it < 100; ++it) hipLaunchKernelGGL(kern, dim3(256), dim3(NTHREAD), 0, 0, dp, IN, OUT, rows, dx, ypp, dhh, dhl);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("  ping-pong ablation %-52s %.1f us\n", name, ms * 10.f);
    };
    timepp(k_fwd_pp<8, 1, 0>, "none");
    timepp(k_fwd_pp<8, 1, 64>, "LEAN vector phase (correct numbers)");
    timepp(k_fwd_pp<8, 1, 64 + 16>, "LEAN vector phase, teams in phase");
    timepp(k_fwd_pp<8, 1, 128>, "TRANSPOSED image (ds_write_b64 / ds_read_b64_tr_b16, correct numbers)");
    timepp(k_fwd_pp<8, 1, 128 + 16>, "TRANSPOSED image, teams in phase");
    timepp(k_fwd_pp<8, 1, 128 + 1024>, "TRANSPOSED image, TEAM-LOCAL barriers");
    timepp(k_fwd_pp<8, 1, 1024>, "original image, TEAM-LOCAL barriers");
    timepp(k_fwd_pp<8, 1, 128 + 256>, "TRANSPOSED image, vector phase at s_setprio 3");
    timepp(k_fwd_pp<8, 1, 128 + 512>, "TRANSPOSED image, matrix phase at s_setprio 3");
    timepp(k_fwd_pp<8, 1, 128 + 2>, "TRANSPOSED image, no exp");
    timepp(k_fwd_pp<8, 1, 128 + 4>, "TRANSPOSED image, no image stores (old path stores skipped)");
    timepp(k_fwd_pp<8, 1, 128 + 8>, "TRANSPOSED image, no output reduction");
    timepp(k_fwd_pp<8, 1, 128 + 32>, "TRANSPOSED image, lo from registers only");
    timepp(k_fwd_pp<8, 1, 16>, "teams IN PHASE (lock-step control)");
    timepp(k_fwd_pp<8, 1, 1>, "one k-block of the matrix block instead of 8");
    timepp(k_fwd_pp<8, 1, 2>, "no exp (ELU = identity)");
    timepp(k_fwd_pp<8, 1, 4>, "no image stores");
    timepp(k_fwd_pp<8, 1, 8>, "no output reduction");
    timepp(k_fwd_pp<8, 1, 32>, "lo halves from registers only (wrong numbers)");
    timepp(k_fwd_pp<8, 1, 14>, "matrix block only (no exp, stores, reduction)");
    timepp(k_fwd_pp<8, 1, 15>, "nothing (one k-block, no exp, stores, reduction)");
    return 0;
}
