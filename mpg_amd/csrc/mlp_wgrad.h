// Weight-gradient workgroup body shared by the single-network and the multi-network launches.
#pragma once
#include "mlp_launch.h"

namespace mlp {

struct WgradArgs {
    int in_dim, out_dim, rows, groups_per_chunk;
    XSpec x;
    const float *h1, *h2, *dz1, *dz2, *dz3;
    float* slabs;
};

template <int IN>
__device__ __forceinline__ float x_value(const XSpec& x, long gr, int i) {
    return i < x.d0 ? x.x0[gr * x.ld0 + i] * x.scale[i] : x.x1[gr * x.ld1 + (i - x.d0)];
}

template <int IN, int OU>
constexpr int wgrad_nq() { return 2 * IN + 4 + 2 * OU + OU; }      // thin quantities per lane

// sl: column slice (hidden columns [32 sl, 32 sl + 32)); chunk: which run of row groups; sRed: NWAVE*NQ*64 floats of LDS
template <int IN, int OU>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, const int sl, const int chunk, float* sRed) {
    constexpr int NQ = wgrad_nq<IN, OU>();
    const Lane L;
    const int tid = threadIdx.x;
    const long ngroups = (a.rows + GROUP - 1) / GROUP;
    const long g0 = (long)chunk * a.groups_per_chunk;
    const long g1 = (g0 + a.groups_per_chunk < ngroups) ? g0 + a.groups_per_chunk : ngroups;
    f32x4 acc[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) acc[u][0] = acc[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    float gW1[2][IN], gb1[2] = {0.f, 0.f}, gb2[2] = {0.f, 0.f}, gW3[2][OU], gb3[OU];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int i = 0; i < IN; ++i) gW1[t][i] = 0.f;
#pragma unroll
        for (int o = 0; o < OU; ++o) gW3[t][o] = 0.f;
    }
#pragma unroll
    for (int o = 0; o < OU; ++o) gb3[o] = 0.f;
    const f32x4* H1 = reinterpret_cast<const f32x4*>(a.h1);
    const f32x4* H2 = reinterpret_cast<const f32x4*>(a.h2);
    const f32x4* DZ1 = reinterpret_cast<const f32x4*>(a.dz1);
    const f32x4* DZ2 = reinterpret_cast<const f32x4*>(a.dz2);
    // ---- thin pieces (dW1, db1, db2, dW3, db3): the chunk's groups are dealt round-robin to the 8 waves.  The loads
    //      of a wave's first thin group are issued before the MFMA loop and consumed after it, so their latency is
    //      hidden; later groups (long chunks, e.g. NADP's 26*B rows) are prefetched one ahead. ----
    struct Thin {
        f32x4 d10, d11, d20, d21, h20, h21;
        float d3[4][OU], x[4][IN];
    };
    auto thin_load = [&](long g, Thin& t) {
        t.d10 = DZ1[(g * 16 + 2 * sl) * 64 + L.lane]; t.d11 = DZ1[(g * 16 + 2 * sl + 1) * 64 + L.lane];
        t.d20 = DZ2[(g * 16 + 2 * sl) * 64 + L.lane]; t.d21 = DZ2[(g * 16 + 2 * sl + 1) * 64 + L.lane];
        t.h20 = H2[(g * 16 + 2 * sl) * 64 + L.lane]; t.h21 = H2[(g * 16 + 2 * sl + 1) * 64 + L.lane];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long gr = g * GROUP + L.row(j);
            const bool live = gr < a.rows;
#pragma unroll
            for (int o = 0; o < OU; ++o) t.d3[j][o] = live ? a.dz3[gr * OU + o] : 0.f;
#pragma unroll
            for (int i = 0; i < IN; ++i) t.x[j][i] = live ? x_value<IN>(a.x, gr, i) : 0.f;
        }
    };
    auto thin_accumulate = [&](const Thin& t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            gb1[0] += t.d10[j]; gb1[1] += t.d11[j];
            gb2[0] += t.d20[j]; gb2[1] += t.d21[j];
#pragma unroll
            for (int i = 0; i < IN; ++i) {
                gW1[0][i] = fmaf(t.x[j][i], t.d10[j], gW1[0][i]);
                gW1[1][i] = fmaf(t.x[j][i], t.d11[j], gW1[1][i]);
            }
#pragma unroll
            for (int o = 0; o < OU; ++o) {
                gW3[0][o] = fmaf(t.h20[j], t.d3[j][o], gW3[0][o]);
                gW3[1][o] = fmaf(t.h21[j], t.d3[j][o], gW3[1][o]);
                if (L.c == 0) gb3[o] += t.d3[j][o];
            }
        }
    };
    Thin tcur;
    long tg = g0 + L.wave;
    const bool has_thin = tg < g1;
    if (has_thin) thin_load(tg, tcur);

    // ---- dW2 on the matrix pipe, fragments of group g+1 in flight while group g multiplies ----
    f32x4 nb0, nb1, na0, na1;
    nb0 = DZ2[(g0 * 16 + 2 * sl) * 64 + L.lane]; nb1 = DZ2[(g0 * 16 + 2 * sl + 1) * 64 + L.lane];
    na0 = H1[(g0 * 16 + 2 * L.wave) * 64 + L.lane]; na1 = H1[(g0 * 16 + 2 * L.wave + 1) * 64 + L.lane];
    for (long g = g0; g < g1; ++g) {
        const f32x4 b0 = nb0, b1 = nb1, a0 = na0, a1 = na1;
        if (g + 1 < g1) {
            nb0 = DZ2[((g + 1) * 16 + 2 * sl) * 64 + L.lane]; nb1 = DZ2[((g + 1) * 16 + 2 * sl + 1) * 64 + L.lane];
            na0 = H1[((g + 1) * 16 + 2 * L.wave) * 64 + L.lane]; na1 = H1[((g + 1) * 16 + 2 * L.wave + 1) * 64 + L.lane];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // the float4's 4 entries are 4 k-steps (k = batch row)
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
        }
    }
    if (has_thin) {
        for (;;) {
            const long tnext = tg + NWAVE;
            Thin tn;
            if (tnext < g1) thin_load(tnext, tn);
            thin_accumulate(tcur);
            if (tnext >= g1) break;
            tcur = tn;
            tg = tnext;
        }
    }
    // ---- this workgroup's part of the chunk slab ----
    float* slab = a.slabs + (size_t)chunk * net_size(a.in_dim, a.out_dim);
    float* sW1 = slab;
    float* sb1 = sW1 + a.in_dim * H;
    float* sW2 = sb1 + H;
    float* sb2 = sW2 + H * H;
    float* sW3 = sb2 + H;
    float* sb3 = sW3 + H * a.out_dim;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                sW2[(16 * (2 * L.wave + u) + 4 * L.rg + j) * H + 32 * sl + 16 * t + L.c] = acc[u][t][j];
    // thin pieces: sum over the 8 waves and the 4 row quads through LDS in a fixed order
    {
        float* dst = sRed + (L.wave * NQ) * 64 + L.lane;
        int q = 0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int i = 0; i < IN; ++i) dst[(q++) * 64] = gW1[t][i];
            dst[(q++) * 64] = gb1[t];
            dst[(q++) * 64] = gb2[t];
#pragma unroll
            for (int o = 0; o < OU; ++o) dst[(q++) * 64] = gW3[t][o];
        }
#pragma unroll
        for (int o = 0; o < OU; ++o) dst[(q++) * 64] = gb3[o];
    }
    __syncthreads();
    for (int item = tid; item < NQ * 16; item += NTHREAD) {
        const int q = item / 16, c = item % 16;
        float sum = 0.f;
        for (int w = 0; w < NWAVE; ++w)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) sum += sRed[(w * NQ + q) * 64 + rg * 16 + c];
        constexpr int PER_T = IN + 2 + OU;
        if (q < 2 * PER_T) {
            const int t = q / PER_T, r = q % PER_T, col = 32 * sl + 16 * t + c;
            if (r < IN) sW1[r * H + col] = sum;
            else if (r == IN) sb1[col] = sum;
            else if (r == IN + 1) sb2[col] = sum;
            else sW3[col * a.out_dim + (r - IN - 2)] = sum;
        } else if (sl == 0) {
            // db3[o]: lanes with c == 0 carried it; summing over c adds exact zeros
            float tot = sum;
            tot += __shfl_xor(tot, 1, 16); tot += __shfl_xor(tot, 2, 16); tot += __shfl_xor(tot, 4, 16); tot += __shfl_xor(tot, 8, 16);
            if (c == 0) sb3[q - 2 * PER_T] = tot;
        }
    }
    // unused output columns of W3 / b3 (the log-std half of the policy head, SURVEY B-5) have zero gradient
    for (int item = tid; item < 32 * (a.out_dim - OU); item += NTHREAD) {
        const int col = 32 * sl + item / (a.out_dim - OU), o = OU + item % (a.out_dim - OU);
        sW3[col * a.out_dim + o] = 0.f;
    }
    if (sl == 0 && tid < a.out_dim - OU) sb3[OU + tid] = 0.f;
}



inline int wgrad_groups_per_chunk(long ngroups) {
    long gp = (ngroups + 31) / 32;   // <= 32 chunk slabs
    return (int)(gp < 1 ? 1 : gp);
}

}  // namespace mlp
