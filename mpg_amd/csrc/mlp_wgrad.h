// Weight-gradient workgroup body shared by the single-network and the multi-network launches.
#pragma once
#include <stdlib.h>

#include "mlp_launch.h"

namespace mlp {

struct WgradArgs {
    int in_dim, out_dim, rows, groups_per_chunk;
    int no_thin;            // only dW2: the thin parts were accumulated elsewhere (reverse sweep, THIN) - their slab entries are zeros
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
// ROLE 0: the workgroup computes its 256 x 32 slice of dW2 AND the thin pieces of its 32 columns; 1: dW2 only; 2: the thin pieces only.
// The thin pieces are a chain of load round trips that runs as a tail behind the matrix loop (~5 us of the bench step's k_wgrad_multi,
// tools/ab_wg_nothin.sh); k_wgrad_multi deals the two roles to different workgroups, the matrix ones with 64-column slices (NT = 4).
// NT: 16-column tiles of DZ2 per workgroup (2: a 32-column slice, 8 workgroups per chunk; 4: a 64-column slice, 4 per chunk - the dW2-only
// role of the launches that carry no thin pieces: every workgroup of a chunk re-reads the chunk's H1 through L2, and halving that traffic
// is worth 18 - 23 % of the kernel, tools/ab_wg_half_a.sh)
template <int IN, int OU, int ROLE = 0, int NT = 2>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, const int sl, const int chunk, float* sRed) {
    static_assert(NT == 2 || ROLE == 1, "wide column slices are built for the dW2-only role");
    constexpr int NQ = wgrad_nq<IN, OU>();
    const Lane L;
    const int tid = threadIdx.x;
    const long ngroups = (a.rows + GROUP - 1) / GROUP;
    const long g0 = (long)chunk * a.groups_per_chunk;
    const long g1 = (g0 + a.groups_per_chunk < ngroups) ? g0 + a.groups_per_chunk : ngroups;
    f32x4 acc[2][NT];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
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
    // ---- thin pieces (dW1, db1, db2, dW3, db3): the chunk's groups are dealt round-robin to the 8 waves and handled
    //      after the matrix loop (requesting a wave's first group ahead of the loop costs 27 registers across it: 35 spills
    //      at the 128-register cap) ----
    struct Thin {
        f32x4 d10, d11, d20, d21, h20, h21;
    };
    auto thin_load = [&](long g, Thin& t) {
        t.d10 = DZ1[(g * 16 + 2 * sl) * 64 + L.lane]; t.d11 = DZ1[(g * 16 + 2 * sl + 1) * 64 + L.lane];
        t.d20 = DZ2[(g * 16 + 2 * sl) * 64 + L.lane]; t.d21 = DZ2[(g * 16 + 2 * sl + 1) * 64 + L.lane];
        t.h20 = H2[(g * 16 + 2 * sl) * 64 + L.lane]; t.h21 = H2[(g * 16 + 2 * sl + 1) * 64 + L.lane];
    };
    // the per-row inputs (x, dz3) of the group are staged through a wave-private corner of the LDS scratch (64-bit
    // global addressing for 4 rows x (IN + OU) scalars per lane would cost ~100 registers and one workgroup of residency)
    // a staged row: [x (8 or 16 floats) | dz3 (OU) | pad] - 12 floats for networks with up to 8 inputs, 20 for the 16-wide ones
    constexpr int XW = xs_of<IN>(), RS = XW + 4, NSLOT = GROUP * RS / 64;
    float* stage = sRed + L.wave * (GROUP * RS);
    // Every lane fills three slots of the staging corner.  Which array a slot comes from (x0 scaled, x1, dz3, or a
    // zero pad) is settled by SELECTING the address, and the loads themselves are unconditional: as three nested
    // branches each slot cost up to three dependent memory round trips (the wait sat inside the branch), ~2 us per
    // row group on the critic jobs, which were then the longest workgroups of the launch.
    struct Staged {
        float v[NSLOT];          // raw loads (consumed by stage_write: the ROLE 2 loop requests a group ahead)
    };
    auto stage_load = [&](long g, Staged& st) {
#pragma unroll
        for (int u = 0; u < NSLOT; ++u) {
            const int e = L.lane + 64 * u, row = e / RS, i = e % RS;
            const long gr = g * GROUP + row;
            const bool from_x0 = i < a.in_dim && i < a.x.d0, from_x1 = i < a.in_dim && !from_x0, from_d3 = i >= XW && i - XW < OU;
            const bool on = gr < a.rows && (from_x0 || from_x1 || from_d3);
            const float* p = a.dz3;                                      // any valid address for the slots that stay zero
            if (on) p = from_x0 ? a.x.x0 + gr * a.x.ld0 + i : (from_x1 ? a.x.x1 + gr * a.x.ld1 + (i - a.x.d0) : a.dz3 + gr * OU + (i - XW));
            st.v[u] = *p;
        }
    };
    auto stage_write = [&](long g, const Staged& st) {
#pragma unroll
        for (int u = 0; u < NSLOT; ++u) {
            const int e = L.lane + 64 * u, row = e / RS, i = e % RS;
            const long gr = g * GROUP + row;
            const bool from_x0 = i < a.in_dim && i < a.x.d0, from_x1 = i < a.in_dim && !from_x0, from_d3 = i >= XW && i - XW < OU;
            const bool on = gr < a.rows && (from_x0 || from_x1 || from_d3);
            float sc = a.x.scale[i < 24 ? i : 0];
            if (!from_x0) sc = 1.f;
            stage[L.lane + 64 * u] = on ? st.v[u] * sc : 0.f;
        }
    };
    auto thin_accumulate = [&](long g, const Thin& t, const Staged& st) {
        stage_write(g, st);
        __builtin_amdgcn_wave_barrier();      // same wave writes and reads: LDS is in order within a wave
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* rowp = stage + L.row(j) * RS;
            float d3[OU], x[IN];
#pragma unroll
            for (int i = 0; i < IN; ++i) x[i] = rowp[i];
#pragma unroll
            for (int o = 0; o < OU; ++o) d3[o] = rowp[XW + o];
            // Every multiply-add of this block is ONE hand-written single-precision instruction.  Left to the compiler
            // they become packed v_pk_fma_f32 / v_pk_add_f32 (op_sel forms), and in k_wgrad_multi single products were then
            // lost nondeterministically: first the (row 13, even i) terms of dW1 in ~85 % of launches, after dW1 was
            // unpacked the t = 1 half of dW3 in ~5 % (found by the repeated-launch determinism
            // check, not reproduced in isolation by archive/proto/pk_hazard.hip; a build without packed fp32 is clean).
#define MPG_FMAC(acc, a_, b_) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(a_), "v"(b_))
#define MPG_ADD(acc, a_) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(a_))
            MPG_ADD(gb1[0], t.d10[j]); MPG_ADD(gb1[1], t.d11[j]);
            MPG_ADD(gb2[0], t.d20[j]); MPG_ADD(gb2[1], t.d21[j]);
#pragma unroll
            for (int i = 0; i < IN; ++i) {
                MPG_FMAC(gW1[0][i], x[i], t.d10[j]);
                MPG_FMAC(gW1[1][i], x[i], t.d11[j]);
            }
#pragma unroll
            for (int o = 0; o < OU; ++o) {
                MPG_FMAC(gW3[0][o], t.h20[j], d3[o]);
                MPG_FMAC(gW3[1][o], t.h21[j], d3[o]);
                if (L.c == 0) gb3[o] += d3[o];
            }
#undef MPG_FMAC
#undef MPG_ADD
        }
        __builtin_amdgcn_wave_barrier();
    };
#ifdef MPG_SPLIT
    float dz_scale = 1.f;
    if constexpr (ROLE != 2) {
    // Scale of the DZ2 operand of the split-fp16 product: a power of two taken from the chunk's own data, 2^(4 - e) with e the
    // exponent of max |dL/dz3| over the chunk's rows (|dz2| <= out * max|dz3| * max|W3|, so the scaled operand stays below
    // 32 * max|W3| < 65504 inside the parameter envelope).  A fixed scale of ~B (the 1/B of the loss mean) left per-sample
    // gradients of 1e-3 .. 1e-4 - small TD errors, late-training policy gradients - with 3 .. 0 bits in their fp16 lo halves.
    // The accumulators are scaled back per workgroup, so chunks are free to differ.
    {
        float mx = 0.f;
        const long r0 = g0 * GROUP * OU, r1 = ((g1 * GROUP < (long)a.rows) ? g1 * GROUP : (long)a.rows) * OU;
        for (long i = r0 + tid; i < r1; i += NTHREAD) mx = fmaxf(mx, fabsf(a.dz3[i]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (L.lane == 0) sRed[L.wave] = mx;
        __syncthreads();
        float m = sRed[0];
#pragma unroll
        for (int w = 1; w < NWAVE; ++w) m = fmaxf(m, sRed[w]);
        __syncthreads();                                   // sRed is the B tile / the staging corners from here on
        int e = __builtin_amdgcn_frexp_expf(m);            // 0 for m == 0
        e = e < -100 ? -100 : (e > 100 ? 100 : e);
        dz_scale = ldexpf(A_SCALE, -e);
    }
    }
#endif
    long tg = g0 + L.wave;
    const bool has_thin = tg < g1 && !a.no_thin && ROLE != 1;

    // ---- dW2 on the matrix pipe.  A operand: this wave's 32 rows of dW2 = 2 fragments of H1 per row group, straight
    //      from the stash (HBM / Infinity Cache, ~1.5 us away).  B operand: the workgroup's 32-column slice of DZ2 - the
    //      SAME two fragments for all eight waves, so they go through LDS: each wave fetches them for one group of an
    //      8-group tile and every wave reads the tile back (this removes 7/16 of the L2 -> CU traffic). ----
    if constexpr (ROLE != 2) {
#ifdef MPG_SPLIT
    // Split-fp16 form (mlp_core.h): the contraction index is the batch row, 32 rows = TWO row groups per
    // v_mfma_f32_16x16x32_f16: lane (c, rg) supplies rows 4rg..4rg+3 of both groups of a pair - exactly the two float4 it
    // loads from the G16 stash - as hi/lo halves; three MFMAs (hi*hi, hi*lo, lo*hi) per tile pair.  H1 enters as x*16, DZ2 as
    // dz * dz_scale (the chunk's data-dependent power of two, above), undone at the end.
    auto split8 = [](const f32x4& u, const f32x4& v, float sc, f16x8& hi, f16x8& lo) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = u[j] * sc, b = v[j] * sc;
            hi[j] = (_Float16)a; hi[4 + j] = (_Float16)b;
            lo[j] = (_Float16)(a - (float)hi[j]); lo[4 + j] = (_Float16)(b - (float)hi[4 + j]);
        }
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 fa[2][2];                                       // [group of the pair][A tile]: raw floats of the NEXT pair
    auto a_load = [&](long g) {                            // pair (g, g + 1)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const bool ok = g + e < g1;
            fa[e][0] = ok ? H1[((g + e) * 16 + 2 * L.wave) * 64 + L.lane] : zero4;
            fa[e][1] = ok ? H1[((g + e) * 16 + 2 * L.wave + 1) * 64 + L.lane] : zero4;
        }
    };
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2* sB2 = reinterpret_cast<f32x2*>(sRed);          // 8-byte slots: (((pair*2 + tile)*2 + part)*64 + lane)*2 + (group & 1)
    a_load(g0);
    MPG_TL(1);
    for (long tile = g0; tile < g1; tile += NWAVE) {
        const long gb = tile + L.wave;
        f32x4 bt[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) bt[t] = gb < g1 ? DZ2[(gb * 16 + NT * sl + t) * 64 + L.lane] : zero4;
        // this wave's group of the tile, split once for all eight readers
        f32x2 bh[NT], bl[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4& b = bt[t];
            _Float16 h[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = b[j] * dz_scale;
                h[j] = (_Float16)x;
                l[j] = (_Float16)(x - (float)h[j]);
            }
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            bh[t] = __builtin_bit_cast(f32x2, f16x4{h[0], h[1], h[2], h[3]});
            bl[t] = __builtin_bit_cast(f32x2, f16x4{l[0], l[1], l[2], l[3]});
        }
        __syncthreads();                                  // the previous tile has been read by every wave
        {
            const int pr = L.wave >> 1, e = L.wave & 1;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                sB2[(((pr * NT + t) * 2 + 0) * 64 + L.lane) * 2 + e] = bh[t];
                sB2[(((pr * NT + t) * 2 + 1) * 64 + L.lane) * 2 + e] = bl[t];
            }
        }
        __syncthreads();
#pragma unroll
        for (int pr = 0; pr < NWAVE / 2; ++pr) {
            const long g = tile + 2 * pr;
            if (g < g1) {
                f16x8 ah[2], al[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) split8(fa[0][u], fa[1][u], A_SCALE, ah[u], al[u]);
                a_load(g + 2);                             // the next pair (zeros beyond the chunk)
                const f32x4* sB4 = reinterpret_cast<const f32x4*>(sRed);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const f16x8 fbh = __builtin_bit_cast(f16x8, sB4[((pr * NT + t) * 2 + 0) * 64 + L.lane]);
                    const f16x8 fbl = __builtin_bit_cast(f16x8, sB4[((pr * NT + t) * 2 + 1) * 64 + L.lane]);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u], fbh, acc[u][t], 0, 0, 0);
                        acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u], fbl, acc[u][t], 0, 0, 0);
                        acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[u], fbh, acc[u][t], 0, 0, 0);
                    }
                }
            }
        }
    }
    {
        const float un = 1.f / (A_SCALE * dz_scale);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[u][t][j] *= un;
    }
#else
    static_assert(NT == 2, "the exact-fp32 engine keeps 32-column slices");
    constexpr int DEPTH = 2;
    f32x4 fa0[DEPTH], fa1[DEPTH];
    auto a_load = [&](long g, int slot) {
        fa0[slot] = H1[(g * 16 + 2 * L.wave) * 64 + L.lane]; fa1[slot] = H1[(g * 16 + 2 * L.wave + 1) * 64 + L.lane];
    };
    f32x4* sB = reinterpret_cast<f32x4*>(sRed);          // [NWAVE groups][2 fragments][64 lanes], aliases the thin scratch
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (g0 + d < g1) a_load(g0 + d, d);
    for (long tile = g0; tile < g1; tile += NWAVE) {
        const long gb = tile + L.wave;
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
        if (gb < g1) {
            b0 = DZ2[(gb * 16 + 2 * sl) * 64 + L.lane];
            b1 = DZ2[(gb * 16 + 2 * sl + 1) * 64 + L.lane];
        }
        __syncthreads();                                  // the previous tile has been read by every wave
        sB[(L.wave * 2) * 64 + L.lane] = b0;
        sB[(L.wave * 2 + 1) * 64 + L.lane] = b1;
        __syncthreads();
#pragma unroll
        for (int d = 0; d < NWAVE; ++d) {
            const long g = tile + d;
            if (g < g1) {
                const int slot = d & (DEPTH - 1);             // tiles start at multiples of NWAVE from g0: (g - g0) & 1 == d & 1
                const f32x4 fb0 = sB[(d * 2) * 64 + L.lane], fb1 = sB[(d * 2 + 1) * 64 + L.lane];
                const f32x4 a0 = fa0[slot], a1 = fa1[slot];
                if (g + DEPTH < g1) a_load(g + DEPTH, slot);
#pragma unroll
                for (int j = 0; j < 4; ++j) {   // the float4's 4 entries are 4 k-steps (k = batch row)
                    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], fb0[j], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], fb1[j], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], fb0[j], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], fb1[j], acc[1][1], 0, 0, 0);
                }
            }
        }
    }
#endif
    }
    MPG_TL(2);
    __syncthreads();                                      // the staging corners of the thin part alias the B tile
    MPG_TL(3);
    // thin pieces after the matrix loop (their registers are then free): the chunk's groups are dealt round-robin to
    // the 8 waves; the load latency is covered by the other resident waves
    if (has_thin) {
        // (requesting the next group's loads ahead of the sums - tried in the ROLE 2 form, round 4 - costs 55 registers and was slower:
        // k_wgrad_multi 25.4 -> 27.8 us)
        for (; tg < g1; tg += NWAVE) {
            Thin tcur;
            Staged scur;
            thin_load(tg, tcur);
            stage_load(tg, scur);
            thin_accumulate(tg, tcur, scur);
        }
    }
    MPG_TL(4);
    // ---- this workgroup's part of the chunk slab ----
    float* slab = a.slabs + (size_t)chunk * net_size(a.in_dim, a.out_dim);
    float* sW1 = slab;
    float* sb1 = sW1 + a.in_dim * H;
    float* sW2 = sb1 + H;
    float* sb2 = sW2 + H * H;
    float* sW3 = sb2 + H;
    float* sb3 = sW3 + H * a.out_dim;
    if constexpr (ROLE != 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    sW2[(16 * (2 * L.wave + u) + 4 * L.rg + j) * H + 16 * NT * sl + 16 * t + L.c] = acc[u][t][j];
    }
    if constexpr (ROLE == 1) return;                      // (the thin entries of the slab belong to the ROLE 2 workgroup of this slice)
    MPG_TL(5);
    // thin pieces: sum over the 8 waves and the 4 row quads through LDS in a fixed order
    __syncthreads();   // the staging corners used above alias this scratch
    MPG_TL(6);
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
            if (r < IN) { if (r < a.in_dim) sW1[r * H + col] = sum; }       // (the 16-wide instantiation: rows beyond in_dim do not exist)
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



// 1-D grid of 8*nch workgroups -> (chunk, column slice).  Workgroups are dealt round-robin to the 8 XCDs, each with
// its own L2; all 8 column slices of a chunk re-read the same H1 tiles, so they are mapped to the SAME XCD and to
// consecutive dispatch slots there: the re-reads then hit that XCD's L2 instead of going to HBM 8 times.
// (speed only - any mapping is correct.)
__device__ __forceinline__ void wgrad_map(int b, int nch, int& chunk, int& sl) {
    if ((nch & 7) == 0) {
        const int xcd = b & 7, k = b >> 3;
        chunk = xcd + 8 * (k >> 3);
        sl = k & 7;
    } else {
        chunk = b >> 3;
        sl = b & 7;
    }
}

// the same for the 64-column slices (four workgroups per chunk)
__device__ __forceinline__ void wgrad_map4(int b, int nch, int& chunk, int& sl) {
    if ((nch & 7) == 0) {
        const int xcd = b & 7, k = b >> 3;
        chunk = xcd + 8 * (k >> 2);
        sl = k & 3;
    } else {
        chunk = b >> 2;
        sl = b & 3;
    }
}

// Chunking of a network's weight-gradient job: every chunk is 8 workgroups (one per 32-column slice) that leave one slab of
// partial sums for the reduction.  16 chunks per network: the bench step's three jobs are then 384 workgroups, all resident at
// once (512 slots), and 13 MB of slabs.  Measured on the bench step (tools/ab_repeat.sh, k_wgrad_multi / whole step):
// 32 per network 28 us / 0.264 ms (768 workgroups: a second, mostly empty round; 26 MB of slabs), 16: 24.4 / 0.2555,
// 8 + 8 + 48 by job size: 28.5 / 0.258, 8 per network: 29 / 0.259.  The launch as a whole moves ~430 MB of stash through
// the Infinity Cache in that time; short workgroups mostly wait in the same queues as the long ones.
// A launch that carries ONE network's job (NADP, TD3, the fine-grained entry points) wants more, shorter chunks: 64 (512
// workgroups) - C3 NADP B = 8192: 1.06 ms per gradient step with 16, 0.87 with 32, 0.84 with 64; C4 TD3 B = 65 536: 1.40 /
// 1.25 / 1.245 (tools/bench_configs.py).
#ifndef MPG_WGRAD_MAX_CHUNKS
#define MPG_WGRAD_MAX_CHUNKS 16
#endif
#ifndef MPG_WGRAD_MAX_CHUNKS_SINGLE
#define MPG_WGRAD_MAX_CHUNKS_SINGLE 64
#endif
#ifndef MPG_WGRAD_MAX_CHUNKS_W2
#define MPG_WGRAD_MAX_CHUNKS_W2 128      // the dW2-only launch with 64-column slices: 4 workgroups per chunk, 512 in all
#endif
constexpr int WGRAD_MAX_CHUNKS = MPG_WGRAD_MAX_CHUNKS, WGRAD_MAX_CHUNKS_SINGLE = MPG_WGRAD_MAX_CHUNKS_SINGLE;
// (jobs of 4096 row groups and more - TD3 at B = 65 536 - take at least MPG_WGRAD_W2_MIN_GROUPS groups per chunk: 64 chunks there
// instead of 128 halve the 32 MB of slabs the summation launch reads; C4 0.838 -> 0.830 ms, tools/ab_side.sh.  Smaller jobs keep up
// to 128 chunks: they need the workgroups.)
#ifndef MPG_WGRAD_W2_MIN_GROUPS
#define MPG_WGRAD_W2_MIN_GROUPS 64
#endif
inline int wgrad_groups_per_chunk_w2(long ngroups) {
    long gp = (ngroups + MPG_WGRAD_MAX_CHUNKS_W2 - 1) / MPG_WGRAD_MAX_CHUNKS_W2;
    if (ngroups >= 4096 && gp < MPG_WGRAD_W2_MIN_GROUPS) gp = MPG_WGRAD_W2_MIN_GROUPS;
    return (int)(gp < 1 ? 1 : gp);
}
inline int wgrad_groups_per_chunk(long ngroups, bool single_job = false) {
    const long mc = single_job ? WGRAD_MAX_CHUNKS_SINGLE : WGRAD_MAX_CHUNKS;
    long gp = (ngroups + mc - 1) / mc;
    return (int)(gp < 1 ? 1 : gp);
}

}  // namespace mlp
