// Prototype (round 5): FOUR waves x 512 registers.
//   workgroup = 256 threads = 4 waves = ONE wave per SIMD (__launch_bounds__(256, 1): the wave may use the whole 512-entry unified
//   register file of its SIMD lane: 256 architectural VGPRs + 256 AccVGPRs).  Wave q owns 64 hidden columns (4 tiles of 16); the
//   hi AND lo halves of its 256 x 64 slice of W2 are register-stationary (2 x 128 registers - the compiler is free to keep them in
//   AccVGPRs: gfx950's MFMA takes A / B operands from either file), everything else has 256 registers to itself: no spills,
//   each A-operand row of the activation image feeds 4 column tiles instead of 2 (half the ds_reads per SIMD), 4-wave barriers,
//   no young/old arbitration between SIMD partners, and with TILE-MAJOR order tile t's epilogue can run under tile t + 1's MFMAs.
// Harness: the chain-free forward pass of archive/proto/pingpong (65 536 rows of an 8 -> 256 -> 256 -> 1 network, 16 row groups per
// workgroup), in two modes:
//   FREE   groups are independent (next inputs prefetched under the matrix block) - the judge's pass criterion: < 30 us per launch
//          against 35.3 for the product's lock-step pairs;
//   CHAIN  the next group's input depends on this group's OUTPUT through LDS and a third barrier, like a rollout step (minus the
//          model) - the per-group time here x 26 is what the sweeps would see.
// Build + run:  bash archive/proto/agpr4/run.sh   (on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstring>
#include "mlp_core.h"

using namespace mlp;

// ---- the shipped geometry as the reference: k_forward's pairs (FREE) and forward_group one group at a time (CHAIN) -----------
template <int IN, int OU, bool CHAIN>
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
    const int tid = threadIdx.x;
    if constexpr (CHAIN) {
        const float b3 = net.b3[0];
        float carry = 0.f;                      // the output lanes' previous output (enters the next input with weight 0)
        for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
            if (tid < GROUP * XS) {
                const int row = tid / XS, i = tid % XS;
                const long gr = g * GROUP + row;
                sX[tid] = ((gr < rows && i < in_dim) ? x[gr * in_dim + i] : 0.f) + 0.f * carry;
            }
            lds_barrier();
            float h1[2][4], h2[2][4];
            forward_group<IN, OU>(sX, sA, sPart, L, w2, r, h1, h2);
            if (tid < GROUP * XS) {
                const int row = tid / XS;
                carry = out_preact_tree(sPart, b3, row, 0);
                const long gr = g * GROUP + row;
                if (tid % XS == 0 && gr < rows) y[gr * OU] = carry;
            }
        }
    } else {
        const long nunits = (ngroups + 1) / 2;
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
}

// ---- four waves x 512 registers ----------------------------------------------------------------------------------------------
constexpr int TW = 4;                 // waves per workgroup
constexpr int NT = 4;                 // 16-column tiles per wave
constexpr int NTH4 = 64 * TW;

// OPT bits: 1 tile-major matrix block (A fragments of all 8 k-blocks preloaded: 64 registers), 2 transposed activation image,
//           4 s_setprio 1 during the matrix block; AB (timing only, wrong numbers): 16 one k-block, 32 no exps, 64 no image stores
template <int IN, int OU, bool CHAIN, int OPT>
__global__ void __launch_bounds__(NTH4, 1) k_fwd_a4(const float* params, int in_dim, int out_dim, int rows, const float* x, float* y,
                                                    const float* pk_hi, const float* pk_lo) {
    __shared__ __attribute__((aligned(16))) float sAimg[A_IMG];
    __shared__ __attribute__((aligned(16))) float sX[GROUP * XS];
    __shared__ float sPart[2][TW * GROUP * MAXOUT];
    const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6, c = lane & 15, rg = lane >> 4;
    const Net net = make_net(params, in_dim, out_dim);
    float whi[128], wlo[128];         // [(kb * NT + t) * 4 + r]: packed pair (k0, k0 + 1), k0 = 32 kb + 8 rg + 2 r, column 64 q + 16 t + c
    {
        const f32x4* ph = reinterpret_cast<const f32x4*>(pk_hi) + (q * 8 * NT) * 64 + lane;
        const f32x4* pl = reinterpret_cast<const f32x4*>(pk_lo) + (q * 8 * NT) * 64 + lane;
#pragma unroll
        for (int v = 0; v < 8 * NT; ++v) {
            const f32x4 h = ph[v * 64], l = pl[v * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e) { whi[v * 4 + e] = h[e]; wlo[v * 4 + e] = l[e]; }
        }
    }
    float w1p[2][NT], b1[NT], b2[NT], w3[NT][OU];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = 64 * q + 16 * t + c;
#pragma unroll
        for (int s = 0; s < 2; ++s) w1p[s][t] = (4 * s + rg) < in_dim ? net.W1[(4 * s + rg) * H + col] : 0.f;
        b1[t] = net.b1[col];
        b2[t] = net.b2[col];
#pragma unroll
        for (int o = 0; o < OU; ++o) w3[t][o] = net.W3[col * out_dim + o];
    }
    const float b3v = net.b3[0];
    auto hfrag = [&](int kb, int t) { const int v = (kb * NT + t) * 4; return __builtin_bit_cast(f16x8, f32x4{whi[v], whi[v + 1], whi[v + 2], whi[v + 3]}); };
    auto lfrag = [&](int kb, int t) { const int v = (kb * NT + t) * 4; return __builtin_bit_cast(f16x8, f32x4{wlo[v], wlo[v + 1], wlo[v + 2], wlo[v + 3]}); };
    _Float16* sH = reinterpret_cast<_Float16*>(sAimg);
    const _Float16* bh = sH + rg * PLANE_H + c * ROW_H;
    const _Float16* bl = bh + IMG_H;
    const bool odd = c & 1;
    const int row0 = 4 * rg + (odd ? 2 : 0);
    const long ngroups = (rows + GROUP - 1) / GROUP;
    constexpr bool TILE_MAJOR = OPT & 1, TR = OPT & 2;
    constexpr int NKB = (OPT & 16) ? 1 : 8;
    auto x_regs = [&](long g, float (&xa)[2]) {        // FREE mode: this lane's layer-1 A operand straight from global memory
        const long gr = g * GROUP + c;
#pragma unroll
        for (int s = 0; s < 2; ++s) xa[s] = (g < ngroups && gr < rows && 4 * s + rg < in_dim) ? x[gr * in_dim + 4 * s + rg] : 0.f;
    };
    float xa[2];
    float carry = 0.f;
    if (!CHAIN) x_regs(blockIdx.x, xa);
    int par = 0;
    for (long g = blockIdx.x; g < ngroups; g += gridDim.x, par ^= 1) {
        if constexpr (CHAIN) {
            if (tid < GROUP * XS) {
                const int row = tid / XS, i = tid % XS;
                const long gr = g * GROUP + row;
                sX[tid] = ((gr < rows && i < in_dim) ? x[gr * in_dim + i] : 0.f) + 0.f * carry;
            }
            lds_barrier();
#pragma unroll
            for (int s = 0; s < 2; ++s) xa[s] = sX[c * XS + 4 * s + rg];
        }
        // ---- layer 1 + ELU + fp16 split + image store ----
        {
            f32x4 z[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) z[t] = f32x4{b1[t], b1[t], b1[t], b1[t]};
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < NT; ++t) z[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], w1p[s][t], z[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float h[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    h[j] = (OPT & 32) ? z[t][j] : __builtin_amdgcn_fmed3f(z[t][j], __builtin_amdgcn_exp2f(z[t][j] * 1.4426950408889634f) - 1.f, 0.f);
                if constexpr (TR) {
                    unsigned h01, l01, h23, l23;
                    split2_mix(h[0], h[1], A_SCALE, h01, l01);
                    split2_mix(h[2], h[3], A_SCALE, h23, l23);
                    const int byte = tr_byte(tr_slot(64 * q + 16 * t + c), rg);
                    if (!(OPT & 64)) {
                        *reinterpret_cast<u32x2*>(reinterpret_cast<char*>(sAimg) + byte) = u32x2{h01, h23};
                        *reinterpret_cast<u32x2*>(reinterpret_cast<char*>(sAimg) + TR_IMG_BYTES + byte) = u32x2{l01, l23};
                    } else asm volatile("" :: "v"(h01), "v"(l01), "v"(h23), "v"(l23));
                } else {
                    float pn[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) pn[j] = dpp_mov<0xB1>(h[j]);
                    const int k = 64 * q + 16 * t + (c & ~1);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const float xx = odd ? pn[2 + u] : h[u], yy = odd ? h[2 + u] : pn[u];
                        float hi, lo;
                        split_pack2(xx * A_SCALE, yy * A_SCALE, hi, lo);
                        if (OPT & 64) asm volatile("" :: "v"(hi), "v"(lo));
                        else {
                            *reinterpret_cast<float*>(sH + h_index(row0 + u, k)) = hi;
                            *reinterpret_cast<float*>(sH + IMG_H + h_index(row0 + u, k)) = lo;
                        }
                    }
                }
            }
        }
        lds_barrier();
        if (!CHAIN) x_regs(g + gridDim.x, xa);                  // the next group's inputs travel under the matrix block
        // ---- matrix block ----
        if constexpr ((OPT & 4) != 0) __builtin_amdgcn_s_setprio(1);
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto a_frag = [&](int kb, f16x8& ah, f16x8& al) {
            if constexpr (TR) {
                typedef __attribute__((address_space(3))) s16x4* lds_p;
                const char* img = reinterpret_cast<const char*>(sAimg);
                const int qq = (lane >> 2) & 3, pp = lane & 3;
                const int s0 = 32 * kb + 16 * (rg >> 1) + 4 * (rg & 1) + qq, s1 = s0 + 8;
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + tr_byte(s0, pp)));
                const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + tr_byte(s1, pp)));
                const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + TR_IMG_BYTES + tr_byte(s0, pp)));
                const s16x4 b1v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + TR_IMG_BYTES + tr_byte(s1, pp)));
                ah = __builtin_bit_cast(f16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
                al = __builtin_bit_cast(f16x8, __builtin_shufflevector(b0, b1v, 0, 1, 2, 3, 4, 5, 6, 7));
            } else {
                ah = *reinterpret_cast<const f16x8*>(bh + 8 * kb);
                al = *reinterpret_cast<const f16x8*>(bl + 8 * kb);
            }
        };
        float p[OU][4];
#pragma unroll
        for (int o = 0; o < OU; ++o)
#pragma unroll
            for (int j = 0; j < 4; ++j) p[o][j] = 0.f;
        auto epilogue_tile = [&](int t) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = fmaf(acc[t][j], 1.f / (W_SCALE * A_SCALE), b2[t]);
                const float h2 = (OPT & 32) ? a : __builtin_amdgcn_fmed3f(a, __builtin_amdgcn_exp2f(a * 1.4426950408889634f) - 1.f, 0.f);
#pragma unroll
                for (int o = 0; o < OU; ++o) p[o][j] = fmaf(h2, w3[t][o], p[o][j]);
            }
        };
        if constexpr (TILE_MAJOR) {
            f16x8 ah[8], al[8];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) a_frag(kb, ah[kb], al[kb]);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kb], hfrag(kb, t), acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kb], lfrag(kb, t), acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[kb], hfrag(kb, t), acc[t], 0, 0, 0);
                }
                epilogue_tile(t);
            }
        } else {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                f16x8 ah, al;
                a_frag(kb, ah, al);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, hfrag(kb, t), acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, lfrag(kb, t), acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, hfrag(kb, t), acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) epilogue_tile(t);
        }
        if constexpr ((OPT & 4) != 0) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int o = 0; o < OU; ++o)
#pragma unroll
            for (int j = 0; j < 4; ++j) p[o][j] = row_allreduce16(p[o][j]);
        if (c == 0) {
#pragma unroll
            for (int o = 0; o < OU; ++o)
#pragma unroll
                for (int j = 0; j < 4; ++j) sPart[par][(q * GROUP + 4 * rg + j) * MAXOUT + o] = p[o][j];
        }
        lds_barrier();
        // ---- output (sPart is double-buffered by group parity: the next group's partials cannot overtake these reads) ----
        if (CHAIN ? tid < GROUP * XS : tid < GROUP * OU) {
            const int row = CHAIN ? tid / XS : tid / OU;
            float zz = b3v;
#pragma unroll
            for (int w = 0; w < TW; ++w) zz += sPart[par][(w * GROUP + row) * MAXOUT];
            carry = zz;
            const long gr = g * GROUP + row;
            if ((CHAIN ? tid % XS == 0 : true) && gr < rows) y[gr * OU] = zz;
        }
    }
}

int main() {
    const int IN = 8, OUT = 1, rows = 65536;
    const int np = net_size(IN, OUT);
    std::vector<float> hp(np), hx((size_t)rows * IN);
    srand(1);
    auto rnd = [] { return (rand() / (float)RAND_MAX) * 2.f - 1.f; };
    for (int i = 0; i < np; ++i) hp[i] = rnd() * 0.1f;
    for (auto& v : hx) v = rnd();
    float *dp, *dx, *y8, *ya;
    hipMalloc(&dp, np * 4); hipMalloc(&dx, hx.size() * 4); hipMalloc(&y8, rows * 4); hipMalloc(&ya, rows * 4);
    hipMemcpy(dp, hp.data(), np * 4, hipMemcpyHostToDevice);
    hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
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
    auto time8 = [&](auto kern, int rr, const char* name) {
        for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(kern, dim3(256), dim3(NTHREAD), 0, 0, dp, IN, OUT, rr, dx, y8);
        hipEventRecord(e0);
        for (int it = 0; it < 100; ++it) hipLaunchKernelGGL(kern, dim3(256), dim3(NTHREAD), 0, 0, dp, IN, OUT, rr, dx, y8);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("rows %6d  %-64s %6.1f us  (%.0f ns per group and CU)\n", rr, name, ms * 10.f, ms * 1e4 / (rr / 16 / 256.0));
        return ms * 10.f;
    };
    auto timea = [&](auto kern, int rr, const char* name, bool check) {
        hipMemset(ya, 0, rows * 4);
        for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(kern, dim3(256), dim3(NTH4), 0, 0, dp, IN, OUT, rr, dx, ya, dhh, dhl);
        hipEventRecord(e0);
        for (int it = 0; it < 100; ++it) hipLaunchKernelGGL(kern, dim3(256), dim3(NTH4), 0, 0, dp, IN, OUT, rr, dx, ya, dhh, dhl);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double md = -1;
        if (check) {
            std::vector<float> a(rr), b(rr);
            hipMemcpy(a.data(), y8, rr * 4, hipMemcpyDeviceToHost);
            hipMemcpy(b.data(), ya, rr * 4, hipMemcpyDeviceToHost);
            md = 0;
            for (int i = 0; i < rr; ++i) md = fmax(md, fabs(a[i] - b[i]));
        }
        printf("rows %6d  %-64s %6.1f us  (%.0f ns per group and CU)  max |diff| vs shipped %.2e  err=%s\n", rr, name, ms * 10.f,
               ms * 1e4 / (rr / 16 / 256.0), md, hipGetErrorString(hipGetLastError()));
        return ms * 10.f;
    };
    for (int rr : {65536, 4096}) {
        printf("---- FREE (independent groups) ----\n");
        time8(k_fwd8<8, 1, false>, rr, "shipped: 8 waves, lock-step pairs");
        timea(k_fwd_a4<8, 1, false, 0>, rr, "4 waves x 512 regs, k-block-major", true);
        timea(k_fwd_a4<8, 1, false, 1>, rr, "4 waves x 512 regs, TILE-major (epilogue under next tile)", true);
        timea(k_fwd_a4<8, 1, false, 2>, rr, "4 waves x 512 regs, k-block-major, transposed image", true);
        timea(k_fwd_a4<8, 1, false, 3>, rr, "4 waves x 512 regs, TILE-major, transposed image", true);
        timea(k_fwd_a4<8, 1, false, 3 + 4>, rr, "   + s_setprio 1 in the matrix block", true);
        printf("---- CHAIN (next input depends on this output: a rollout step without the model) ----\n");
        time8(k_fwd8<8, 1, true>, rr, "shipped: 8 waves, forward_group");
        timea(k_fwd_a4<8, 1, true, 0>, rr, "4 waves x 512 regs, k-block-major", true);
        timea(k_fwd_a4<8, 1, true, 1>, rr, "4 waves x 512 regs, TILE-major", true);
        timea(k_fwd_a4<8, 1, true, 2>, rr, "4 waves x 512 regs, k-block-major, transposed image", true);
        timea(k_fwd_a4<8, 1, true, 3>, rr, "4 waves x 512 regs, TILE-major, transposed image", true);
    }
    printf("---- ablations of the CHAIN form (timing only, wrong numbers), 65 536 rows ----\n");
    timea(k_fwd_a4<8, 1, true, 3 + 16>, rows, "TILE-major, transposed: one k-block of the matrix block", false);
    timea(k_fwd_a4<8, 1, true, 3 + 32>, rows, "TILE-major, transposed: no exps", false);
    timea(k_fwd_a4<8, 1, true, 3 + 64>, rows, "TILE-major, transposed: no image stores", false);
    timea(k_fwd_a4<8, 1, true, 3 + 16 + 32 + 64>, rows, "TILE-major, transposed: all three", false);
    return 0;
}
