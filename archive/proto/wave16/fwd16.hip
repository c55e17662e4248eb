// Prototype (round 3, not shipped): the forward pass of the engine with SIXTEEN waves per workgroup - one 16-column tile and 64
// stationary weight registers per wave, four waves per SIMD - against the shipped geometry (eight waves, two tiles, 128 weight
// registers, two waves per SIMD), on the same rows and weights.  Question: does a SIMD with four in-order waves to choose from
// get through a row group faster than one with two?  Build + run:  bash archive/proto/wave16/run.sh   (on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "mlp_core.h"

using namespace mlp;

template <int IN, int OU>
__global__ void __launch_bounds__(NTHREAD, 2) k_fwd8(const float* params, int in_dim, int out_dim, int rows, const float* x, float* y) {
    __shared__ __attribute__((aligned(16))) float smem[A_IMG + GROUP * XS + NWAVE * GROUP * MAXOUT];
    float* sA = smem;
    float* sX = sA + A_IMG;
    float* sPart = sX + GROUP * XS;
    const Lane L;
    const Net net = make_net(params, in_dim, out_dim);
    float w2[128];
    SmallRegs<IN, OU> r;
    load_small<IN, OU>(net, L, r);
    load_w2_fwd(net.W2, L, w2);
    const long ngroups = (rows + GROUP - 1) / GROUP;
    const int tid = threadIdx.x;
    for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        if (tid < GROUP * XS) {
            const int row = tid / XS, i = tid % XS;
            const long gr = g * GROUP + row;
            sX[tid] = (gr < rows && i < in_dim) ? x[gr * in_dim + i] : 0.f;
        }
        lds_barrier();
        float h1[2][4], h2[2][4];
        forward_group<IN, OU>(sX, sA, sPart, L, w2, r, h1, h2);
        if (tid < GROUP * OU) {
            const int row = tid / OU, o = tid % OU;
            const long gr = g * GROUP + row;
            if (gr < rows) y[gr * OU + o] = out_preact(sPart, net.b3[o], row, o);
        }
    }
}

constexpr int NW16 = 16, NT16 = 1024;

// AB: ablations (timing only, wrong numbers): 1 one k-block instead of 8, 2 no exps, 4 no image stores, 8 no output reduction,
// 16 no barriers
template <int IN, int OU, int AB = 0>
__global__ void __launch_bounds__(NT16, 1) k_fwd16(const float* params, int in_dim, int out_dim, int rows, const float* x, float* y) {
    __shared__ __attribute__((aligned(16))) float sA[A_IMG];
    __shared__ float sX[GROUP * XS];
    __shared__ float sPart[NW16 * GROUP * MAXOUT];
    const int tid = threadIdx.x, lane = tid & 63, T = tid >> 6, c = lane & 15, rg = lane >> 4;
    const Net net = make_net(params, in_dim, out_dim);
    const int col = 16 * T + c;
    float w[64];          // w[4 v + r], v = kb * 2 + part: the B fragments of this tile
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int k0 = 32 * kb + 8 * rg + 2 * r4;
            split_pack2(w_scaled(net.W2[k0 * H + col]), w_scaled(net.W2[(k0 + 1) * H + col]), w[4 * (kb * 2) + r4], w[4 * (kb * 2 + 1) + r4]);
        }
    float w1p[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) w1p[q] = (4 * q + rg) < in_dim ? net.W1[(4 * q + rg) * H + col] : 0.f;
    const float b1 = net.b1[col], b2 = net.b2[col];
    float w3[OU];
#pragma unroll
    for (int o = 0; o < OU; ++o) w3[o] = net.W3[col * out_dim + o];
    auto frag = [&](int v) { return __builtin_bit_cast(f16x8, f32x4{w[4 * v], w[4 * v + 1], w[4 * v + 2], w[4 * v + 3]}); };
    _Float16* sH = reinterpret_cast<_Float16*>(sA);
    const bool odd = c & 1;
    const int row0 = 4 * rg + (odd ? 2 : 0);
    const int kst = 16 * T + (c & ~1);
    const _Float16* bh = sH + rg * PLANE_H + c * ROW_H;
    const _Float16* bl = bh + IMG_H;
    const long ngroups = (rows + GROUP - 1) / GROUP;
    for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        if (tid < GROUP * XS) {
            const int row = tid / XS, i = tid % XS;
            const long gr = g * GROUP + row;
            sX[tid] = (gr < rows && i < in_dim) ? x[gr * in_dim + i] : 0.f;
        }
        if (!(AB & 16)) lds_barrier();
        f32x4 z = {b1, b1, b1, b1};
#pragma unroll
        for (int q = 0; q < 2; ++q) z = __builtin_amdgcn_mfma_f32_16x16x4f32(sX[c * XS + 4 * q + rg], w1p[q], z, 0, 0, 0);
        float h1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) h1[j] = (AB & 2) ? z[j] : __builtin_amdgcn_fmed3f(z[j], __builtin_amdgcn_exp2f(z[j] * 1.4426950408889634f) - 1.f, 0.f);
        {
            float p[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) p[j] = dpp_mov<0xB1>(h1[j]);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float xx = odd ? p[2 + u] : h1[u], yy = odd ? h1[2 + u] : p[u];
                float hi, lo;
                split_pack2(xx * A_SCALE, yy * A_SCALE, hi, lo);
                if (AB & 4) asm volatile("" :: "v"(hi), "v"(lo));
                else {
                    *reinterpret_cast<float*>(sH + h_index(row0 + u, kst)) = hi;
                    *reinterpret_cast<float*>(sH + IMG_H + h_index(row0 + u, kst)) = lo;
                }
            }
        }
        if (!(AB & 16)) lds_barrier();
        f32x4 ma = {0.f, 0.f, 0.f, 0.f}, mb = ma;
#pragma unroll
        for (int kb = 0; kb < ((AB & 1) ? 1 : 8); ++kb) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(bh + 8 * kb), al = *reinterpret_cast<const f16x8*>(bl + 8 * kb);
            ma = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, frag(kb * 2), ma, 0, 0, 0);
            mb = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, frag(kb * 2 + 1), mb, 0, 0, 0);
            mb = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, frag(kb * 2), mb, 0, 0, 0);
        }
        float h2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = fmaf(ma[j] + mb[j], 1.f / (W_SCALE * A_SCALE), b2);
            h2[j] = (AB & 2) ? a : __builtin_amdgcn_fmed3f(a, __builtin_amdgcn_exp2f(a * 1.4426950408889634f) - 1.f, 0.f);
        }
        float p[OU][4];
#pragma unroll
        for (int o = 0; o < OU; ++o)
#pragma unroll
            for (int j = 0; j < 4; ++j) p[o][j] = (AB & 8) ? h2[j] * w3[o] : row_allreduce16(h2[j] * w3[o]);
        if (c == 0) {
#pragma unroll
            for (int o = 0; o < OU; ++o)
#pragma unroll
                for (int j = 0; j < 4; ++j) sPart[(T * GROUP + 4 * rg + j) * MAXOUT + o] = p[o][j];
        }
        if (!(AB & 16)) lds_barrier();
        if (tid < GROUP * OU) {
            const int row = tid / OU, o = tid % OU;
            const long gr = g * GROUP + row;
            float zz = net.b3[o];
#pragma unroll
            for (int w16 = 0; w16 < NW16; ++w16) zz += sPart[(w16 * GROUP + row) * MAXOUT + o];
            if (gr < rows) y[gr * OU + o] = zz;
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
    float *dp, *dx, *y8, *y16;
    hipMalloc(&dp, np * 4); hipMalloc(&dx, hx.size() * 4); hipMalloc(&y8, rows * 4); hipMalloc(&y16, rows * 4);
    hipMemcpy(dp, hp.data(), np * 4, hipMemcpyHostToDevice);
    hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rr : {65536, 8192, 4096}) {
        float t8 = 0, t16 = 0;
        for (int which = 0; which < 2; ++which) {
            for (int it = 0; it < 20; ++it) {
                if (which == 0) hipLaunchKernelGGL((k_fwd8<8, 1>), dim3(256), dim3(NTHREAD), 0, 0, dp, IN, OUT, rr, dx, y8);
                else hipLaunchKernelGGL((k_fwd16<8, 1>), dim3(256), dim3(NT16), 0, 0, dp, IN, OUT, rr, dx, y16);
            }
            hipEventRecord(e0);
            for (int it = 0; it < 100; ++it) {
                if (which == 0) hipLaunchKernelGGL((k_fwd8<8, 1>), dim3(256), dim3(NTHREAD), 0, 0, dp, IN, OUT, rr, dx, y8);
                else hipLaunchKernelGGL((k_fwd16<8, 1>), dim3(256), dim3(NT16), 0, 0, dp, IN, OUT, rr, dx, y16);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            (which == 0 ? t8 : t16) = ms * 10.f;
        }
        std::vector<float> a(rr), b(rr);
        hipMemcpy(a.data(), y8, rr * 4, hipMemcpyDeviceToHost);
        hipMemcpy(b.data(), y16, rr * 4, hipMemcpyDeviceToHost);
        double md = 0, mx = 0;
        for (int i = 0; i < rr; ++i) { md = fmax(md, fabs(a[i] - b[i])); mx = fmax(mx, fabs(a[i])); }
        printf("rows %6d (%2d groups / workgroup): 8 waves x 2 tiles %.1f us   16 waves x 1 tile %.1f us   max |diff| %.2e (max |y| %.2f)\n", rr,
               rr / 16 / 256, t8, t16, md, mx);
    }
    auto time16 = [&](auto kern, const char* name) {
        for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(kern, dim3(256), dim3(NT16), 0, 0, dp, IN, OUT, rows, dx, y16);
        hipEventRecord(e0);
        for (int it = 0; it < 100; ++it) hipLaunchKernelGGL(kern, dim3(256), dim3(NT16), 0, 0, dp, IN, OUT, rows, dx, y16);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("  16-wave ablation %-44s %.1f us\n", name, ms * 10.f);
    };
    time16(k_fwd16<8, 1, 0>, "none");
    time16(k_fwd16<8, 1, 1>, "one k-block of the matrix block instead of 8");
    time16(k_fwd16<8, 1, 2>, "no exp (ELU = identity)");
    time16(k_fwd16<8, 1, 4>, "no image stores");
    time16(k_fwd16<8, 1, 8>, "no output reduction");
    time16(k_fwd16<8, 1, 16>, "no barriers");
    time16(k_fwd16<8, 1, 31>, "all of the above");
    time16(k_fwd16<8, 1, 30>, "all but the matrix block");
    return 0;
}
