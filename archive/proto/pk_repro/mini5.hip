// Which other instructions with a set op_sel bit misbehave beside the straddling MFMA loop?  Same setting as mini3.hip (explicit
// registers, checked against scalar instructions on the same operands), six candidates per step:
//   1 v_pk_fma_f32 op_sel:[0,1,0]   (the known one: low lane takes the HIGH register of src1)
//   2 v_pk_fma_f32 op_sel:[1,0,0]   (... of src0)          3 v_pk_fma_f32 op_sel:[0,0,1]   (... of src2)
//   4 v_pk_mul_f32 op_sel:[0,1]      5 v_pk_add_f32 op_sel:[0,1]
//   6 v_pk_fma_f32 op_sel_hi:[1,0,1] (control: HIGH lane takes the LOW register of src1)
// run by mini5_align.sh, which places the neighbour's MFMA loop head at 28 mod 32.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(512, 4) k_mini5(int steps, int mfma_iters, unsigned long long* bad, float* sink) {
    const int lane = threadIdx.x & 63;
    if (blockIdx.x >= gridDim.x / 2) {
        f16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * (lane + j)); b[j] = (_Float16)(0.02f * (lane - j)); }
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int it = 0; it < mfma_iters; ++it)
#pragma unroll
            for (int u = 0; u < 12; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u & 3], 0, 0, 0);
        sink[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
        return;
    }
    unsigned nlo[6] = {0, 0, 0, 0, 0, 0}, nhi[6] = {0, 0, 0, 0, 0, 0};
    float x0 = 0.37f + 0.013f * lane, x1 = -0.61f + 0.007f * lane, d0 = 1.1e-3f * (1 + (lane & 7)), d1 = -0.9e-3f * (1 + (lane & 15)),
          c0 = 0.25f + 0.001f * lane, c1 = -0.125f - 0.002f * lane;
    for (int s = 0; s < steps; ++s) {
        float p[12];
        asm volatile(
            "v_mov_b32 v16, %12\n v_mov_b32 v17, %13\n v_mov_b32 v10, %14\n v_mov_b32 v11, %15\n v_mov_b32 v20, %16\n v_mov_b32 v21, %17\n"
            "s_nop 4\n"
            "v_pk_fma_f32 v[40:41], v[16:17], v[10:11], v[20:21] op_sel:[0,1,0]\n"
            "v_pk_fma_f32 v[42:43], v[16:17], v[10:11], v[20:21] op_sel:[1,0,0]\n"
            "v_pk_fma_f32 v[44:45], v[16:17], v[10:11], v[20:21] op_sel:[0,0,1]\n"
            "v_pk_mul_f32 v[46:47], v[16:17], v[10:11] op_sel:[0,1]\n"
            "v_pk_add_f32 v[48:49], v[16:17], v[10:11] op_sel:[0,1]\n"
            "v_pk_fma_f32 v[50:51], v[16:17], v[10:11], v[20:21] op_sel_hi:[1,0,1]\n"
            "s_nop 4\n"
            "v_mov_b32 %0, v40\n v_mov_b32 %1, v41\n v_mov_b32 %2, v42\n v_mov_b32 %3, v43\n v_mov_b32 %4, v44\n v_mov_b32 %5, v45\n"
            "v_mov_b32 %6, v46\n v_mov_b32 %7, v47\n v_mov_b32 %8, v48\n v_mov_b32 %9, v49\n v_mov_b32 %10, v50\n v_mov_b32 %11, v51\n"
            : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]), "=&v"(p[7]), "=&v"(p[8]), "=&v"(p[9]),
              "=&v"(p[10]), "=&v"(p[11])
            : "v"(x0), "v"(x1), "v"(d0), "v"(d1), "v"(c0), "v"(c1)
            : "v10", "v11", "v16", "v17", "v20", "v21", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
        // expected (low, high) of each, by scalar instructions: op_sel picks the register of the LOW lane, the high lane takes the high registers
        float r[12];
        auto fma1 = [](float a, float b, float c) { float o; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c)); return o; };
        auto mul1 = [](float a, float b) { float o; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b)); return o; };
        auto add1 = [](float a, float b) { float o; asm volatile("v_add_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b)); return o; };
        r[0] = fma1(x0, d1, c0); r[1] = fma1(x1, d1, c1);
        r[2] = fma1(x1, d0, c0); r[3] = fma1(x1, d1, c1);
        r[4] = fma1(x0, d0, c1); r[5] = fma1(x1, d1, c1);
        r[6] = mul1(x0, d1);     r[7] = mul1(x1, d1);
        r[8] = add1(x0, d1);     r[9] = add1(x1, d1);
        r[10] = fma1(x0, d0, c0); r[11] = fma1(x1, d0, c1);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            nlo[q] += __float_as_uint(p[2 * q]) != __float_as_uint(r[2 * q]);
            nhi[q] += __float_as_uint(p[2 * q + 1]) != __float_as_uint(r[2 * q + 1]);
        }
        x0 = x0 * 0.75f + 0.11f; x1 = 0.3f - x1 * 0.5f; d0 = d0 * 0.5f + 6e-4f; d1 = d1 * 0.5f - 5e-4f; c0 = c0 * 0.5f + 0.1f; c1 = c1 * 0.5f - 0.07f;
    }
    for (int q = 0; q < 6; ++q) {
        if (nlo[q]) atomicAdd(bad + 2 * q, (unsigned long long)nlo[q]);
        if (nhi[q]) atomicAdd(bad + 2 * q + 1, (unsigned long long)nhi[q]);
    }
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 30, steps = argc > 2 ? atoi(argv[2]) : 2000, mfma_iters = argc > 3 ? atoi(argv[3]) : 3000;
    unsigned long long* bad;
    float* sink;
    (void)hipMalloc(&bad, 96);
    (void)hipMalloc(&sink, 512 * 512 * sizeof(float));
    (void)hipMemset(bad, 0, 96);
    hipFunction_t fn = nullptr;
    if (argc > 4) {
        hipModule_t mod;
        if (hipModuleLoad(&mod, argv[4]) != hipSuccess || hipModuleGetFunction(&fn, mod, "_Z7k_mini5iiPyPf") != hipSuccess) { printf("cannot load %s\n", argv[4]); return 2; }
    }
    for (int l = 0; l < launches; ++l) {
        if (fn) {
            int st = steps, mi = mfma_iters;
            void* args[] = {&st, &mi, &bad, &sink};
            if (hipModuleLaunchKernel(fn, 512, 1, 1, 512, 1, 1, 0, 0, args, nullptr) != hipSuccess) { printf("module launch failed\n"); return 2; }
        } else
            hipLaunchKernelGGL(k_mini5, dim3(512), dim3(512), 0, 0, steps, mfma_iters, bad, sink);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
    unsigned long long h[12];
    (void)hipMemcpy(h, bad, 96, hipMemcpyDeviceToHost);
    const char* name[6] = {"v_pk_fma_f32 op_sel:[0,1,0]", "v_pk_fma_f32 op_sel:[1,0,0]", "v_pk_fma_f32 op_sel:[0,0,1]", "v_pk_mul_f32 op_sel:[0,1]", "v_pk_add_f32 op_sel:[0,1]",
                           "v_pk_fma_f32 op_sel_hi:[1,0,1]"};
    for (int q = 0; q < 6; ++q) printf("   %-32s low lane wrong %12llu   high lane wrong %12llu   (of %.3g)\n", name[q], h[2 * q], h[2 * q + 1], (double)launches * 256 * 512 * steps);
    return 0;
}
