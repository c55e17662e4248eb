// Minimal form of the lost packed-FMA products (see README.md): no library code, no memory traffic in the loops.
// Half of the workgroups (the first half of the grid) run chains of v_pk_fma_f32 and, beside each, the same two multiply-adds as
// scalar v_fmac_f32 on the same operands; the other half run back-to-back MFMAs.  Two 512-thread workgroups fit a CU
// (__launch_bounds__(512, 4): 128 VGPRs), so with grid = 2 x 256 every CU holds one workgroup of each kind and every SIMD runs
// packed-FMA waves beside MFMA waves.  A packed half that differs from its scalar twin is counted (low / high halves separately).
//   hipcc --offload-arch=gfx950 -O3 -o mini mini.hip && ./mini [launches] [pk_iters] [mfma_iters]
// -DMFMA_KIND=0: v_mfma_f32_16x16x32_f16 (default)   1: v_mfma_f32_16x16x4_f32   2: v_mfma_f32_32x32x16_f16   3: no MFMA (VALU busy loop)
// -DSEPARATE: the two kinds of workgroup in separate halves of the CHIP instead (blockIdx parity = XCD parity): never co-resident
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#ifndef MFMA_KIND
#define MFMA_KIND 0
#endif

__global__ void __launch_bounds__(512, 4) k_mini(int pk_iters, int mfma_iters, unsigned long long* bad, float* sink) {
    const int lane = threadIdx.x & 63;
#ifdef SEPARATE
    const bool pk_role = (blockIdx.x & 1) == 0;
#else
    const bool pk_role = blockIdx.x < gridDim.x / 2;
#endif
    if (!pk_role) {
        f16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * (lane + j)); b[j] = (_Float16)(0.02f * (lane - j)); }
#if MFMA_KIND == 2
        f32x16 acc[2] = {};
        for (int it = 0; it < mfma_iters; ++it)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u & 1], 0, 0, 0);
        sink[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1];
#elif MFMA_KIND == 3
        float v[4] = {1.f, 2.f, 3.f, 4.f};
        for (int it = 0; it < mfma_iters * 12; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = __builtin_fmaf(v[u], 0.999f, 0.001f);
        sink[blockIdx.x * 512 + threadIdx.x] = v[0] + v[1] + v[2] + v[3];
#else
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int it = 0; it < mfma_iters; ++it)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#if MFMA_KIND == 0
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u], 0, 0, 0);
#else
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a[r], (float)b[u], acc[u], 0, 0, 0);
#endif
                }
        sink[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
#endif
        return;
    }
    // packed role: acc2 (pair) += x2 (pair) * d (splat of the LOW register of the pair d2: op_sel_hi:[1,0,1], the form the compiler
    // made of gW1[t][i..i+1] += x[i..i+1] * dz1[j]); beside it the scalar twins
    unsigned long long nlo = 0, nhi = 0;
    f32x2 x2 = {0.37f + 0.001f * lane, -0.81f + 0.002f * lane}, d2 = {1e-3f * (1 + (lane & 7)), 123.f};
    for (int it = 0; it < pk_iters; ++it) {
        f32x2 acc2 = {0.f, 0.f};
        float r0 = 0.f, r1 = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc2) : "v"(x2), "v"(d2));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r0) : "v"(x2[0]), "v"(d2[0]));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r1) : "v"(x2[1]), "v"(d2[0]));
            // new operands for the next step (exact, cheap, the same for both forms)
            x2[0] = x2[0] * 0.5f + 0.25f;
            x2[1] = 0.75f - x2[1] * 0.5f;
            d2[0] = d2[0] * 0.5f + 5e-4f;
        }
        nlo += __float_as_uint(acc2[0]) != __float_as_uint(r0);
        nhi += __float_as_uint(acc2[1]) != __float_as_uint(r1);
    }
    if (nlo) atomicAdd(bad, nlo);
    if (nhi) atomicAdd(bad + 1, nhi);
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 200, pk_iters = argc > 2 ? atoi(argv[2]) : 2000, mfma_iters = argc > 3 ? atoi(argv[3]) : 4000;
    unsigned long long* bad;
    float* sink;
    (void)hipMalloc(&bad, 16);
    (void)hipMalloc(&sink, 512 * 512 * sizeof(float));
    (void)hipMemset(bad, 0, 16);
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(k_mini, dim3(512), dim3(512), 0, 0, pk_iters, mfma_iters, bad, sink);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
    unsigned long long h[2];
    (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    printf("MFMA_KIND %d%s: %llu low-half and %llu high-half chains differ from their scalar twins (%.3g chains of 32 packed FMAs checked)\n",
           MFMA_KIND,
#ifdef SEPARATE
           " (separate CUs)",
#else
           "",
#endif
           h[0], h[1], (double)launches * 256 * 512 * pk_iters);
    return 0;
}
