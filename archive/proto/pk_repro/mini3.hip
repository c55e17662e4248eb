// The same effect without the compiler's help: explicit registers, explicit instructions, self-checking.
// First half of the grid: per step eight back-to-back v_pk_fma_f32 (the two op_sel splat forms the compiler used for
// gW1[t][i..i+1] += x[i..i+1] * dz_t[j]) on operands made by plain arithmetic, each half checked against a scalar v_fma_f32 of the same
// operands.  Second half of the grid: back-to-back v_mfma_f32_16x16x32_f16 on registers.  512 workgroups of 512 threads, two per CU.
//   hipcc --offload-arch=gfx950 -O3 -o mini3 mini3.hip && ./mini3 [launches] [steps] [mfma_iters]      (-DNEIGHBOR=0: no MFMAs)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#ifndef NEIGHBOR
#define NEIGHBOR 1
#endif
#ifndef MFMA_KIND
#define MFMA_KIND 0      // 0: v_mfma_f32_16x16x32_f16   1: v_mfma_f32_16x16x4_f32   2: v_fma_f32 (no matrix instruction)
#endif

__global__ void __launch_bounds__(512, 4) k_mini3(int steps, int mfma_iters, unsigned long long* bad, float* sink) {
    const int lane = threadIdx.x & 63;
    if (blockIdx.x >= gridDim.x / 2) {
#if NEIGHBOR
        f16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * (lane + j)); b[j] = (_Float16)(0.02f * (lane - j)); }
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int it = 0; it < mfma_iters; ++it)
#pragma unroll
            for (int u = 0; u < 12; ++u) {
#if MFMA_KIND == 1
                acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a[u & 7], (float)b[u & 7], acc[u & 3], 0, 0, 0);
#elif MFMA_KIND == 2      // not a matrix instruction at all: an 8-byte VOP3 in the same loop shape
                acc[u & 3][0] = __builtin_fmaf(acc[u & 3][0], 0.999f, 0.001f * (float)a[0]);
#else
                acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u & 3], 0, 0, 0);
#endif
            }
        sink[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
#endif
        return;
    }
    unsigned long long nlo = 0, nhi = 0;
    unsigned per[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long nzero = 0, nother = 0;          // a deviating low half that is exactly 0 (= its addend: the product is missing) / x * the OTHER register of the pair      // low-half mismatches by instruction of the block
    float x[8], d[4], e[4];
    for (int i = 0; i < 8; ++i) x[i] = 0.37f + 0.013f * (lane + 7 * i);
    for (int j = 0; j < 4; ++j) { d[j] = 1.1e-3f * (1 + ((lane + j) & 7)); e[j] = -0.9e-3f * (1 + ((lane + 3 * j) & 15)); }
    for (int s = 0; s < steps; ++s) {
        float p[16];      // results: pairs (lo, hi)
        asm volatile(
            "v_mov_b32 v16, %16\n v_mov_b32 v17, %17\n v_mov_b32 v18, %18\n v_mov_b32 v19, %19\n"
            "v_mov_b32 v20, %20\n v_mov_b32 v21, %21\n v_mov_b32 v22, %22\n v_mov_b32 v23, %23\n"
            "v_mov_b32 v10, %24\n v_mov_b32 v11, %25\n v_mov_b32 v12, %26\n v_mov_b32 v13, %27\n"
            "v_mov_b32 v6, %28\n v_mov_b32 v7, %29\n v_mov_b32 v8, %30\n v_mov_b32 v9, %31\n"
            "s_nop 4\n"
            "v_pk_fma_f32 v[40:41], v[16:17], v[10:11], 0 op_sel_hi:[1,0,0]\n"      // (x0, x1) * d0
            "v_pk_fma_f32 v[42:43], v[16:17], v[6:7], 0 op_sel_hi:[1,0,0]\n"        // (x0, x1) * e0
            "v_pk_fma_f32 v[44:45], v[18:19], v[10:11], 0 op_sel_hi:[1,0,0]\n"      // (x2, x3) * d0
            "v_pk_fma_f32 v[46:47], v[18:19], v[6:7], 0 op_sel_hi:[1,0,0]\n"        // (x2, x3) * e0
            "v_pk_fma_f32 v[48:49], v[20:21], v[10:11], 0 op_sel:[0,1,0]\n"         // (x4, x5) * d1
            "v_pk_fma_f32 v[50:51], v[20:21], v[6:7], 0 op_sel:[0,1,0]\n"           // (x4, x5) * e1
            "v_pk_fma_f32 v[52:53], v[22:23], v[12:13], 0 op_sel_hi:[1,0,0]\n"      // (x6, x7) * d2
            "v_pk_fma_f32 v[54:55], v[22:23], v[8:9], 0 op_sel:[0,1,0]\n"           // (x6, x7) * e3
            "s_nop 4\n"
            "v_mov_b32 %0, v40\n v_mov_b32 %1, v41\n v_mov_b32 %2, v42\n v_mov_b32 %3, v43\n v_mov_b32 %4, v44\n v_mov_b32 %5, v45\n v_mov_b32 %6, v46\n v_mov_b32 %7, v47\n"
            "v_mov_b32 %8, v48\n v_mov_b32 %9, v49\n v_mov_b32 %10, v50\n v_mov_b32 %11, v51\n v_mov_b32 %12, v52\n v_mov_b32 %13, v53\n v_mov_b32 %14, v54\n v_mov_b32 %15, v55\n"
            : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]), "=&v"(p[7]),
              "=&v"(p[8]), "=&v"(p[9]), "=&v"(p[10]), "=&v"(p[11]), "=&v"(p[12]), "=&v"(p[13]), "=&v"(p[14]), "=&v"(p[15])
            : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]),
              "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3])
            : "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23",
              "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        const float sp[8] = {d[0], e[0], d[0], e[0], d[1], e[1], d[2], e[3]};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int xi = q < 2 ? 0 : (q < 4 ? 2 : (q < 6 ? 4 : 6));
            float r0, r1;
            asm volatile("v_fma_f32 %0, %1, %2, 0" : "=v"(r0) : "v"(x[xi]), "v"(sp[q]));
            asm volatile("v_fma_f32 %0, %1, %2, 0" : "=v"(r1) : "v"(x[xi + 1]), "v"(sp[q]));
            nlo += __float_as_uint(p[2 * q]) != __float_as_uint(r0);
            per[q] += __float_as_uint(p[2 * q]) != __float_as_uint(r0);
            if (__float_as_uint(p[2 * q]) != __float_as_uint(r0)) {
                const float oth[8] = {d[1], e[1], d[1], e[1], d[0], e[0], d[3], e[2]};
                float ro;
                asm volatile("v_fma_f32 %0, %1, %2, 0" : "=v"(ro) : "v"(x[xi]), "v"(oth[q]));
                nzero += p[2 * q] == 0.f;
                nother += __float_as_uint(p[2 * q]) == __float_as_uint(ro);
            }
            nhi += __float_as_uint(p[2 * q + 1]) != __float_as_uint(r1);
        }
        // next step's operands
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = x[i] * 0.75f + 0.11f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { d[j] = d[j] * 0.5f + 6e-4f; e[j] = e[j] * 0.5f - 5e-4f; }
    }
    if (nlo) atomicAdd(bad, nlo);
    if (nhi) atomicAdd(bad + 1, nhi);
    for (int q = 0; q < 8; ++q)
        if (per[q]) atomicAdd(bad + 2 + q, (unsigned long long)per[q]);
    if (nzero) atomicAdd(bad + 10, nzero);
    if (nother) atomicAdd(bad + 11, nother);
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 100, steps = argc > 2 ? atoi(argv[2]) : 2000, mfma_iters = argc > 3 ? atoi(argv[3]) : 3000;
    unsigned long long* bad;
    float* sink;
    (void)hipMalloc(&bad, 96);
    (void)hipMalloc(&sink, 512 * 512 * sizeof(float));
    (void)hipMemset(bad, 0, 96);
    // argv[4]: a code object with k_mini3 assembled from an edited copy of the compiler's assembly (mini3_align.sh places the MFMA loop)
    hipFunction_t fn = nullptr;
    if (argc > 4) {
        hipModule_t mod;
        if (hipModuleLoad(&mod, argv[4]) != hipSuccess || hipModuleGetFunction(&fn, mod, "_Z7k_mini3iiPyPf") != hipSuccess) { printf("cannot load %s\n", argv[4]); return 2; }
    }
    for (int l = 0; l < launches; ++l) {
        if (fn) {
            int st = steps, mi = mfma_iters;
            void* args[] = {&st, &mi, &bad, &sink};
            if (hipModuleLaunchKernel(fn, 512, 1, 1, 512, 1, 1, 0, 0, args, nullptr) != hipSuccess) { printf("module launch failed\n"); return 2; }
        } else
            hipLaunchKernelGGL(k_mini3, dim3(512), dim3(512), 0, 0, steps, mfma_iters, bad, sink);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
    unsigned long long h[12];
    (void)hipMemcpy(h, bad, 96, hipMemcpyDeviceToHost);
    printf("NEIGHBOR=%d MFMA_KIND=%d: %llu low halves and %llu high halves of %.3g packed FMAs differ from the scalar v_fma_f32 of the same operands\n", NEIGHBOR, MFMA_KIND, h[0], h[1],
           (double)launches * 256 * 512 * steps * 8);
    if (h[0]) printf("   low halves by instruction (1-4, 7: op_sel_hi:[1,0,0]; 5, 6, 8: op_sel:[0,1,0]): %llu %llu %llu %llu %llu %llu %llu %llu\n   of these: exactly 0 (the product is missing) %llu, the product with the OTHER register of the src1 pair %llu\n", h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11]);
    return 0;
}
