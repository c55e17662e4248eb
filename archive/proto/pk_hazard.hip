// Does a packed-fp32 FMA lose its low half when another workgroup on the same CU is in an f16-MFMA phase?
// (diagnosis of the round-2 weight-gradient nondeterminism; see archive/proto/README.md)
//   hipcc --offload-arch=gfx950 -O3 -o pk_hazard pk_hazard.hip && ./pk_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(512, 4) k(const float* __restrict__ d_in, const float* __restrict__ x_in, int iters, int mfma_iters,
                                            unsigned long long* bad, float* sink, int mode) {
    __shared__ __attribute__((aligned(16))) float lds[8 * 16 * 12 + 4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool matrix_role = mode == 0 ? (blockIdx.x & 1) : (mode == 1 ? false : true);
    if (matrix_role) {
        f16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * (lane + j)); b[j] = (_Float16)(0.02f * (lane - j)); }
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int it = 0; it < mfma_iters; ++it) {
            f32x4* l4 = reinterpret_cast<f32x4*>(lds + 8 * 16 * 12);
            __syncthreads();
            f32x2* l2 = reinterpret_cast<f32x2*>(l4);
            for (int q = 0; q < 4; ++q) l2[((((wave >> 1) * 4 + q) * 64 + lane) * 2 + (wave & 1)) & 2047] = f32x2{acc[0][q], acc[1][q]};
            __syncthreads();
            f32x4 t = {0, 0, 0, 0};
            for (int q = 0; q < 16; ++q) t += l4[(q * 64 + lane) & 1023];
            // the split-fp16 operand preparation of the real matrix loop (packed multiplies, conversions, packed subtracts)
            f32x4 u0 = t, u1 = acc[1];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float p0 = u0[jj] * 16.f, p1 = u1[jj] * 16.f;
                a[jj] = (_Float16)p0; a[4 + jj] = (_Float16)p1;
                b[jj] = (_Float16)(p0 - (float)a[jj]); b[4 + jj] = (_Float16)(p1 - (float)a[4 + jj]);
            }
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u], 0, 0, 0);
        }
        sink[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
        return;
    }
    // thin role: per iteration stage 16 rows x 12 floats through a wave-private LDS corner, then the compiler's own schedule of
    // the weight-gradient kernel's thin loop (physical registers, LDS returns in flight while the packed FMAs execute)
    float* stage = lds + wave * 192;
    const int c = lane & 15, rg = lane >> 4;
    unsigned long long nbad = 0;
    for (int it = 0; it < iters; ++it) {
        const long g = ((long)blockIdx.x * iters + it) * 8 + wave;
        const f32x4 d0 = reinterpret_cast<const f32x4*>(d_in)[(g * 2) % 65536 * 64 + lane];
        const f32x4 d1 = reinterpret_cast<const f32x4*>(d_in)[(g * 2 + 1) % 65536 * 64 + lane];
        for (int e = lane; e < 192; e += 64) stage[e] = x_in[(g * 192 + e) % (1 << 22)];
        __builtin_amdgcn_wave_barrier();
        const unsigned addr = (unsigned)(size_t)(stage + 4 * rg * 12) & 0xffff;   // LDS byte address of row 4rg
        f32x2 pk[2][4];
        asm volatile(
            "v_mov_b32 v80, %8\n v_mov_b32 v81, %9\n v_mov_b32 v82, %10\n v_mov_b32 v83, %11\n"
            "v_mov_b32 v84, %12\n v_mov_b32 v85, %13\n v_mov_b32 v86, %14\n v_mov_b32 v87, %15\n"
            "v_mov_b32 v64, 0\n v_mov_b32 v65, 0\n v_mov_b32 v66, 0\n v_mov_b32 v67, 0\n v_mov_b32 v68, 0\n v_mov_b32 v69, 0\n v_mov_b32 v70, 0\n v_mov_b32 v71, 0\n"
            "v_mov_b32 v72, 0\n v_mov_b32 v73, 0\n v_mov_b32 v74, 0\n v_mov_b32 v75, 0\n v_mov_b32 v76, 0\n v_mov_b32 v77, 0\n v_mov_b32 v78, 0\n v_mov_b32 v79, 0\n"
            "ds_read_b128 v[100:103], %16\n"               // row 0 x0..3
            "ds_read_b128 v[104:107], %16 offset:16\n"     // row 0 x4..7
            "ds_read_b128 v[108:111], %16 offset:48\n"     // row 1 x0..3
            "s_waitcnt lgkmcnt(0)\n"
            "ds_read_b128 v[112:115], %16 offset:96\n"     // row 2 x0..3
            "v_pk_fma_f32 v[74:75], v[102:103], v[84:85], v[74:75] op_sel_hi:[1,0,1]\n"
            "v_pk_fma_f32 v[66:67], v[102:103], v[80:81], v[66:67] op_sel_hi:[1,0,1]\n"
            "v_pk_fma_f32 v[72:73], v[100:101], v[84:85], v[72:73] op_sel_hi:[1,0,1]\n"
            "v_pk_fma_f32 v[64:65], v[100:101], v[80:81], v[64:65] op_sel_hi:[1,0,1]\n"
            "v_pk_fma_f32 v[74:75], v[110:111], v[84:85], v[74:75] op_sel:[0,1,0]\n"
            "v_pk_fma_f32 v[66:67], v[110:111], v[80:81], v[66:67] op_sel:[0,1,0]\n"
            "v_pk_fma_f32 v[72:73], v[108:109], v[84:85], v[72:73] op_sel:[0,1,0]\n"
            "v_pk_fma_f32 v[64:65], v[108:109], v[80:81], v[64:65] op_sel:[0,1,0]\n"
            "ds_read_b128 v[116:119], %16 offset:144\n"    // row 3 x0..3
            "s_waitcnt lgkmcnt(1)\n"
            "v_pk_fma_f32 v[66:67], v[114:115], v[82:83], v[66:67] op_sel_hi:[1,0,1]\n"
            "v_pk_fma_f32 v[74:75], v[114:115], v[86:87], v[74:75] op_sel_hi:[1,0,1]\n"
            "ds_read_b128 v[120:123], %16 offset:64\n"     // row 1 x4..7
            "v_pk_fma_f32 v[64:65], v[112:113], v[82:83], v[64:65] op_sel_hi:[1,0,1]\n"
            "v_pk_fma_f32 v[72:73], v[112:113], v[86:87], v[72:73] op_sel_hi:[1,0,1]\n"
            "ds_read_b128 v[112:115], %16 offset:112\n"    // row 2 x4..7
            "ds_read_b128 v[124:127], %16 offset:160\n"    // row 3 x4..7
            "v_pk_fma_f32 v[76:77], v[104:105], v[84:85], v[76:77] op_sel_hi:[1,0,1]\n"
            "v_pk_fma_f32 v[68:69], v[104:105], v[80:81], v[68:69] op_sel_hi:[1,0,1]\n"
            "v_pk_fma_f32 v[78:79], v[106:107], v[84:85], v[78:79] op_sel_hi:[1,0,1]\n"
            "v_pk_fma_f32 v[70:71], v[106:107], v[80:81], v[70:71] op_sel_hi:[1,0,1]\n"
            "s_waitcnt lgkmcnt(2)\n"
            "v_pk_fma_f32 v[68:69], v[120:121], v[80:81], v[68:69] op_sel:[0,1,0]\n"
            "v_pk_fma_f32 v[70:71], v[122:123], v[80:81], v[70:71] op_sel:[0,1,0]\n"
            "s_waitcnt lgkmcnt(1)\n"
            "v_pk_fma_f32 v[68:69], v[112:113], v[82:83], v[68:69] op_sel_hi:[1,0,1]\n"
            "v_pk_fma_f32 v[70:71], v[114:115], v[82:83], v[70:71] op_sel_hi:[1,0,1]\n"
            "v_pk_fma_f32 v[76:77], v[120:121], v[84:85], v[76:77] op_sel:[0,1,0]\n"
            "v_pk_fma_f32 v[78:79], v[122:123], v[84:85], v[78:79] op_sel:[0,1,0]\n"
            "v_pk_fma_f32 v[64:65], v[116:117], v[82:83], v[64:65] op_sel:[0,1,0]\n"
            "v_pk_fma_f32 v[72:73], v[116:117], v[86:87], v[72:73] op_sel:[0,1,0]\n"
            "v_pk_fma_f32 v[66:67], v[118:119], v[82:83], v[66:67] op_sel:[0,1,0]\n"
            "s_waitcnt lgkmcnt(0)\n"
            "v_pk_fma_f32 v[68:69], v[124:125], v[82:83], v[68:69] op_sel:[0,1,0]\n"
            "v_pk_fma_f32 v[70:71], v[126:127], v[82:83], v[70:71] op_sel:[0,1,0]\n"
            "v_pk_fma_f32 v[76:77], v[112:113], v[86:87], v[76:77] op_sel_hi:[1,0,1]\n"
            "v_pk_fma_f32 v[78:79], v[114:115], v[86:87], v[78:79] op_sel_hi:[1,0,1]\n"
            "v_pk_fma_f32 v[74:75], v[118:119], v[86:87], v[74:75] op_sel:[0,1,0]\n"
            "v_pk_fma_f32 v[76:77], v[124:125], v[86:87], v[76:77] op_sel:[0,1,0]\n"
            "v_pk_fma_f32 v[78:79], v[126:127], v[86:87], v[78:79] op_sel:[0,1,0]\n"
            "v_mov_b32 %0, v64\n v_mov_b32 %1, v66\n v_mov_b32 %2, v68\n v_mov_b32 %3, v70\n v_mov_b32 %4, v72\n v_mov_b32 %5, v74\n v_mov_b32 %6, v76\n v_mov_b32 %7, v78\n"
            : "=&v"(pk[0][0][0]), "=&v"(pk[0][1][0]), "=&v"(pk[0][2][0]), "=&v"(pk[0][3][0]), "=&v"(pk[1][0][0]), "=&v"(pk[1][1][0]), "=&v"(pk[1][2][0]), "=&v"(pk[1][3][0])
            : "v"(d0[0]), "v"(d0[1]), "v"(d0[2]), "v"(d0[3]), "v"(d1[0]), "v"(d1[1]), "v"(d1[2]), "v"(d1[3]), "v"(addr)
            : "memory", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83",
              "v84", "v85", "v86", "v87", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",
              "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
        // scalar reference of the LOW halves (even i) straight from LDS
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float r = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) r = __builtin_fmaf(stage[(4 * rg + j) * 12 + 2 * i], t == 0 ? d0[j] : d1[j], r);
                if (__float_as_uint(pk[t][i][0]) != __float_as_uint(r)) nbad += 1;
            }
        __builtin_amdgcn_wave_barrier();
    }
    if (nbad) atomicAdd(bad, nbad);
    if (c == 99) sink[0] = 1.f;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 64, mfma_iters = argc > 2 ? atoi(argv[2]) : 400, launches = argc > 3 ? atoi(argv[3]) : 50;
    float *d, *x, *sink; unsigned long long* bad;
    const size_t nd = 65536ull * 64 * 4 + 1024, nx = 1 << 22;
    hipMalloc(&d, nd * 4); hipMalloc(&x, nx * 4); hipMalloc(&sink, 4096 * 512 * 4); hipMalloc(&bad, 8);
    float* h = (float*)malloc((nd > nx ? nd : nx) * 4);
    srand(1);
    for (size_t i = 0; i < nd; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 1e-3f;
    hipMemcpy(d, h, nd * 4, hipMemcpyHostToDevice);
    for (size_t i = 0; i < nx; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 3.f;
    hipMemcpy(x, h, nx * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {      // 0: mixed roles (co-resident matrix and thin workgroups)   1: thin only
        hipMemset(bad, 0, 8);
        for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(k, dim3(2048), dim3(512), 0, 0, d, x, iters, mfma_iters, bad, sink, mode);
        hipDeviceSynchronize();
        unsigned long long hb = 0; hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
        printf("mode %d (%s): low-half mismatches %llu, high-half mismatches %llu   (%.3g packed FMAs checked)\n", mode,
               mode == 0 ? "matrix + thin workgroups co-resident" : "thin workgroups only", hb & 0xffffffffull, hb >> 32,
               (double)launches * (mode == 0 ? 1024 : 2048) * 512 * iters * 32);
    }
    return 0;
}
