// Self-contained reduction of the weight-gradient kernel's thin loop (no library code): which ingredients beside the packed FMAs are
// needed to lose products when a co-resident workgroup issues MFMAs?   See README.md; built and run by mini2.sh.
//   first half of the grid: "thin" workgroups - per wave and step: two float4 of dz from global memory (K_GLOBAL) or arithmetic, 16 rows x
//   12 floats of x staged through a wave-private LDS corner (K_LDS) or arithmetic, an exec-masked side sum (K_EXEC), then
//   gW1[t][i..i+1] += x[i..i+1] * dz_t[j] as packed FMAs (PK=1, __builtin_elementwise_fma on float2) or scalar fmaf (PK=0);
//   second half: back-to-back v_mfma_f32_16x16x32_f16 on registers (NEIGHBOR=1) or plain FMAs (NEIGHBOR=0).
// Every launch repeats the same work; launches whose output differs bit for bit from the first launch's are counted.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#ifndef PK
#define PK 1
#endif
#ifndef K_LDS
#define K_LDS 1
#endif
#ifndef K_GLOBAL
#define K_GLOBAL 1
#endif
#ifndef K_EXEC
#define K_EXEC 1
#endif
#ifndef K_COPY
#define K_COPY 0
#endif
#ifndef K_NOP
#define K_NOP 0
#endif
#ifndef K_STORE_NOP
#define K_STORE_NOP 0
#endif
#ifndef K_LDSWAIT
#define K_LDSWAIT 0
#endif
#ifndef NEIGHBOR
#define NEIGHBOR 1
#endif
#ifndef STEPS
#define STEPS 6
#endif
constexpr int IN = 8, RS = 12;

__global__ void __launch_bounds__(512, 4) k_mini2(const float* __restrict__ dz, const float* __restrict__ xs, int neighbor_iters, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float lds[8 * 16 * RS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, rg = lane >> 4;
    const int half = gridDim.x / 2;
    if ((int)blockIdx.x >= half) {
#if NEIGHBOR == 1
        f16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * (lane + j)); b[j] = (_Float16)(0.02f * (lane - j)); }
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int it = 0; it < neighbor_iters; ++it)
#pragma unroll
            for (int u = 0; u < 12; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u & 3], 0, 0, 0);
        const float keep = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
#else
        float v[4] = {1.f, 2.f, 3.f, 4.f};
        for (int it = 0; it < neighbor_iters * 12; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = __builtin_fmaf(v[u], 0.999f, 0.001f);
        const float keep = v[0] + v[1] + v[2] + v[3];
#endif
        if (keep == 12345.678f) out[0] = keep;
        return;
    }
    float* stage = lds + wave * (16 * RS);
    float gW1[2][IN], gb1[2] = {0.f, 0.f}, side = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < IN; ++i) gW1[t][i] = 0.f;
    for (int step = 0; step < STEPS; ++step) {
        const long g = ((long)blockIdx.x * STEPS + step) * 8 + wave;
        f32x4 d0, d1;
#if K_GLOBAL == 1
        d0 = reinterpret_cast<const f32x4*>(dz)[(g * 2) * 64 + lane];
        d1 = reinterpret_cast<const f32x4*>(dz)[(g * 2 + 1) * 64 + lane];
#elif K_GLOBAL == 2   // computed, but with full-width mantissas like the random data (an integer hash)
        for (int j = 0; j < 4; ++j) {
            unsigned h0 = (unsigned)(g * 64 + lane) * 2654435761u + j * 40503u, h1 = h0 * 2246822519u + 374761393u;
            h0 ^= h0 >> 15; h1 ^= h1 >> 13;
            d0[j] = ((float)(h0 & 0xffffff) - 8388608.f) * 1.1920929e-10f;
            d1[j] = ((float)(h1 & 0xffffff) - 8388608.f) * 1.1920929e-10f;
        }
#else
        for (int j = 0; j < 4; ++j) { d0[j] = 1e-3f * (float)((g * 7 + lane * 3 + j) % 29 - 14); d1[j] = 1e-3f * (float)((g * 5 + lane + j * 3) % 31 - 15); }
#endif
#if K_COPY      // the packed FMAs read COPIES of the loaded values (v_mov_b32), not the load's destination registers
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            asm volatile("v_mov_b32 %0, %1" : "=v"(d0[j]) : "v"(d0[j]));
            asm volatile("v_mov_b32 %0, %1" : "=v"(d1[j]) : "v"(d1[j]));
        }
#endif
#if K_NOP       // wait states between the load's return and the first consumer
        asm volatile("s_waitcnt vmcnt(0)\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7" ::: "memory");
#endif
        float xr[4][IN];
#if K_LDS
        for (int u = 0; u < 3; ++u) stage[lane + 64 * u] = xs[g * (16 * RS) + lane + 64 * u];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < IN; ++i) xr[j][i] = stage[(4 * rg + j) * RS + i];
#if K_LDSWAIT   // every LDS return has landed before the first packed FMA (instead of the compiler's staggered s_waitcnt lgkmcnt(7 .. 0))
#pragma unroll
        for (int j = 0; j < 4; ++j)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xr[j][0]), "+v"(xr[j][1]), "+v"(xr[j][2]), "+v"(xr[j][3]), "+v"(xr[j][4]), "+v"(xr[j][5]), "+v"(xr[j][6]), "+v"(xr[j][7])::"memory");
#endif
#else
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < IN; ++i) xr[j][i] = 0.125f * (float)((g + 4 * rg + j + 3 * i) % 17 - 8);
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            gb1[0] += d0[j]; gb1[1] += d1[j];
#if K_EXEC
            if (c == 0) side += xr[j][IN - 1] * d0[j];
#endif
#if PK
#pragma unroll
            for (int i = 0; i + 1 < IN; i += 2) {
                const f32x2 xx = {xr[j][i], xr[j][i + 1]};
                f32x2 a0 = {gW1[0][i], gW1[0][i + 1]}, a1 = {gW1[1][i], gW1[1][i + 1]};
                a0 = __builtin_elementwise_fma(xx, f32x2{d0[j], d0[j]}, a0);
                a1 = __builtin_elementwise_fma(xx, f32x2{d1[j], d1[j]}, a1);
                gW1[0][i] = a0[0]; gW1[0][i + 1] = a0[1]; gW1[1][i] = a1[0]; gW1[1][i + 1] = a1[1];
            }
#else
#pragma unroll
            for (int i = 0; i < IN; ++i) {
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(gW1[0][i]) : "v"(xr[j][i]), "v"(d0[j]));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(gW1[1][i]) : "v"(xr[j][i]), "v"(d1[j]));
            }
#endif
        }
        __builtin_amdgcn_wave_barrier();
    }
#if K_STORE_NOP      // wait states between the last packed FMA and the stores of its results
    asm volatile("s_nop 7\n s_nop 7" ::: "memory");
#endif
    float* o = out + ((size_t)blockIdx.x * 512 + threadIdx.x) * 20;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < IN; ++i) o[t * IN + i] = gW1[t][i];
    o[16] = gb1[0]; o[17] = gb1[1]; o[18] = side; o[19] = 0.f;
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 300, neighbor_iters = argc > 2 ? atoi(argv[2]) : 600;
    const int grid = 512, half = grid / 2;
    const size_t ngroups = (size_t)half * STEPS * 8, ndz = ngroups * 2 * 64 * 4, nx = ngroups * 16 * RS, nout = (size_t)half * 512 * 20;
    std::vector<float> h(ndz > nx ? ndz : nx);
    float *dz, *xs, *out;
    (void)hipMalloc(&dz, ndz * 4); (void)hipMalloc(&xs, nx * 4); (void)hipMalloc(&out, nout * 4);
    srand(3);
    const bool patterned = argc > 3 && atoi(argv[3]) == 1;      // the arithmetic pattern of K_GLOBAL=0, but loaded from memory
    for (size_t i = 0; i < ndz; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 2e-3f;
    if (patterned)
        for (size_t g = 0; g < ngroups; ++g)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    h[((g * 2) * 64 + lane) * 4 + j] = 1e-3f * (float)((long)((g * 7 + lane * 3 + j) % 29) - 14);
                    h[((g * 2 + 1) * 64 + lane) * 4 + j] = 1e-3f * (float)((long)((g * 5 + lane + j * 3) % 31) - 15);
                }
    (void)hipMemcpy(dz, h.data(), ndz * 4, hipMemcpyHostToDevice);
    for (size_t i = 0; i < nx; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    (void)hipMemcpy(xs, h.data(), nx * 4, hipMemcpyHostToDevice);
    // argv[4]: a code object holding k_mini2 (assembled from an edited copy of the compiler's own assembly: asm_bisect.sh) to launch
    // instead of the compiled kernel
    hipFunction_t fn = nullptr;
    if (argc > 4) {
        hipModule_t mod;
        if (hipModuleLoad(&mod, argv[4]) != hipSuccess || hipModuleGetFunction(&fn, mod, "_Z7k_mini2PKfS0_iPf") != hipSuccess) {
            printf("cannot load %s\n", argv[4]);
            return 2;
        }
    }
    std::vector<float> first(nout), cur(nout);
    int bad = 0;
    size_t lo = 0, hi = 0, other = 0, hist[20] = {0};
    for (int l = 0; l < launches; ++l) {
        (void)hipMemset(out, 0xff, nout * 4);
        if (fn) {
            int ni = neighbor_iters;
            void* args[] = {&dz, &xs, &ni, &out};
            if (hipModuleLaunchKernel(fn, grid, 1, 1, 512, 1, 1, 0, 0, args, nullptr) != hipSuccess) { printf("module launch failed\n"); return 2; }
        } else
        hipLaunchKernelGGL(k_mini2, dim3(grid), dim3(512), 0, 0, dz, xs, neighbor_iters, out);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
        (void)hipMemcpy(cur.data(), out, nout * 4, hipMemcpyDeviceToHost);
        if (l == 0) { first = cur; continue; }
        if (memcmp(first.data(), cur.data(), nout * 4) != 0) {
            ++bad;
            for (size_t i = 0; i < nout; ++i)
                if (memcmp(&first[i], &cur[i], 4) != 0) { const int e = (int)(i % 20); ++hist[e]; if (e >= 16) ++other; else if (e & 1) ++hi; else ++lo; }
        }
    }
    printf("%sPK=%d LDS=%d GLOBAL=%d EXEC=%d COPY=%d NOP=%d NEIGHBOR=%d STEPS=%d: %d of %d launches differ from the first; differing accumulators: %zu low halves (even i), %zu high halves, %zu others\n",
           K_LDSWAIT ? "(LDSWAIT) " : K_STORE_NOP ? "(STORE_NOP) " : (patterned ? "(patterned dz) " : ""), PK, K_LDS, K_GLOBAL, K_EXEC, K_COPY, K_NOP, NEIGHBOR, STEPS, bad, launches - 1, lo, hi, other);
#if K_LDS == 1 && K_GLOBAL == 1 && STEPS == 1
    // what do the deviating low halves contain?  Host reference of the LAST launch (fmaf chain, j ascending) and three hypotheses per
    // deviating accumulator: one product missing; one product taken with the OTHER half of the dz pair (d[j ^ 1]); one product taken
    // with the other half of the x pair (x[i ^ 1])
    if (bad && !patterned) {
        std::vector<float> hd(ndz), hx(nx);
        (void)hipMemcpy(hd.data(), dz, ndz * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hx.data(), xs, nx * 4, hipMemcpyDeviceToHost);
        size_t wrong = 0, h_missing = 0, h_dswap = 0, h_xswap = 0, shown = 0, miss_j[16][4] = {{0}};
        for (int b = 0; b < half; ++b)
            for (int tid = 0; tid < 512; ++tid) {
                const int lane = tid & 63, wave = tid >> 6, rg = lane >> 4;
                const size_t g = ((size_t)b * STEPS) * 8 + wave;
                for (int t = 0; t < 2; ++t)
                    for (int i = 0; i < IN; ++i) {
                        const float got = cur[((size_t)b * 512 + tid) * 20 + t * IN + i];
                        auto chain = [&](int miss, int dswap, int xswap) {
                            float r = 0.f;
                            for (int j = 0; j < 4; ++j) {
                                if (j == miss) continue;
                                const float d = hd[((g * 2 + t) * 64 + lane) * 4 + (j == dswap ? (j ^ 1) : j)];
                                const float x = hx[g * (16 * RS) + (4 * rg + j) * RS + (j == xswap ? (i ^ 1) : i)];
                                r = fmaf(x, d, r);
                            }
                            return r;
                        };
                        if (memcmp(&got, &(const float&)chain(-1, -1, -1), 4) == 0) continue;
                        const float ref = chain(-1, -1, -1);
                        ++wrong;
                        bool m = false, ds = false, xw = false;
                        for (int j = 0; j < 4; ++j) {
                            float v = chain(j, -1, -1); if (memcmp(&got, &v, 4) == 0) { m = true; ++miss_j[t * IN + i][j]; }
                            v = chain(-1, j, -1); ds |= memcmp(&got, &v, 4) == 0;
                            v = chain(-1, -1, j); xw |= memcmp(&got, &v, 4) == 0;
                        }
                        h_missing += m; h_dswap += ds; h_xswap += xw;
                        if (shown < 6 && !m && !ds && !xw) { printf("   unexplained: block %d thread %d t %d i %d got %.9g reference %.9g\n", b, tid, t, i, got, ref); ++shown; }
                    }
            }
        printf("   last launch against the host reference: %zu accumulators deviate; explained by one missing product %zu, by one product with the other half of the dz pair %zu, of the x pair %zu\n",
               wrong, h_missing, h_dswap, h_xswap);
        printf("   which product is missing (rows: output slot, columns: j = 0..3):\n");
        for (int e = 0; e < 16; e += 2) printf("      slot %2d: %zu %zu %zu %zu\n", e, miss_j[e][0], miss_j[e][1], miss_j[e][2], miss_j[e][3]);
    }
#endif
    if (bad) {
        printf("   by output slot (gW1[0][0..7] | gW1[1][0..7] | gb1[0] gb1[1] side -):");
        for (int e = 0; e < 20; ++e) printf(" %zu", hist[e]);
        printf("\n");
    }
    return 0;
}
