// Prototype: 16 x 256 x 256 hidden layer with fp16 hi/lo split operands on v_mfma_f32_16x16x32_f16 (3 MFMA terms, fp32
// accumulate) vs the exact fp32 MFMA, numerics against fp64 and time per step.  Geometry = the shipped engine: 512 threads,
// wave w owns columns [32w, 32w+32), C layout (row 4rg+j, col 32w+16t+c).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
constexpr int LDH = 264;            // halves per row of the LDS A images
#ifdef ONEACC
constexpr float WSCALE = 64.f, LOSCALE = 1.f, ASCALE = 16.f;
#else
constexpr float WSCALE = 64.f, LOSCALE = 2048.f, ASCALE = 1.f;
#endif

// packed weights: uint32 index ((wave*8 + kb)*2 + t)*2 + part(0 hi,1 lo))*4 + r)*64 + lane  -> 2 halves: k = 32kb + 8(l>>4) + 2r, +1 ; n = 32w+16t+(l&15)
__global__ void k_pack(const float* __restrict__ W, uint32_t* __restrict__ P) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // 0 .. 8*8*2*2*4*64-1 = 65535
    const int lane = idx & 63, r = (idx >> 6) & 3, part = (idx >> 8) & 1, t = (idx >> 9) & 1, kb = (idx >> 10) & 7, wave = idx >> 13;
    const int n = 32 * wave + 16 * t + (lane & 15), k0 = 32 * kb + 8 * (lane >> 4) + 2 * r;
    _Float16 out[2];
    for (int e = 0; e < 2; ++e) {
        const float w = W[(k0 + e) * 256 + n] * WSCALE;
        const _Float16 hi = (_Float16)w;
        out[e] = part == 0 ? hi : (_Float16)((w - (float)hi) * LOSCALE);
    }
    P[idx] = (uint32_t)__builtin_bit_cast(unsigned short, out[0]) | ((uint32_t)__builtin_bit_cast(unsigned short, out[1]) << 16);
}

__device__ __forceinline__ float dpp_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
}

template <bool SPLIT>
__global__ void __launch_bounds__(512, 2) k_step(const float* __restrict__ W, const uint32_t* __restrict__ P,
                                                 const float* __restrict__ X, float* __restrict__ Y, int steps, int elu) {
    __shared__ __attribute__((aligned(16))) float sA[16 * 280];                  // fp32 path image
    __shared__ __attribute__((aligned(16))) _Float16 sH[2][16 * LDH];           // split path: hi / lo images
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, rg = lane >> 4;
    float w[128];
    uint32_t wp[128];
    if (SPLIT) {
        const uint4* p = reinterpret_cast<const uint4*>(P);
        // lane's 4 consecutive r of (wave,kb,t,part) are 4 uint32 at stride 64 -> gather as 4 dword loads (prototype)
#pragma unroll
        for (int i = 0; i < 128; ++i) wp[i] = P[(size_t)(wave * 128 + i) * 64 + lane];
        (void)p;
    } else {
#pragma unroll
        for (int q = 0; q < 64; ++q)
#pragma unroll
            for (int t = 0; t < 2; ++t) w[2 * q + t] = W[(4 * q + rg) * 256 + 32 * wave + 16 * t + c];
    }
    float h[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) h[t][j] = X[(size_t)blockIdx.x * 4096 + (4 * rg + j) * 256 + 32 * wave + 16 * t + c];
    const bool odd = c & 1;
    for (int s = 0; s < steps; ++s) {
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (SPLIT) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float p[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) p[j] = dpp_xor1(h[t][j]);
                // even lanes write rows j = 0,1 as (own, partner); odd lanes rows j = 2,3 as (partner, own)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float x = ASCALE * (odd ? p[2 + u] : h[t][u]), y = ASCALE * (odd ? h[t][2 + u] : p[u]);
                    const h2 hi = {(_Float16)x, (_Float16)y};
                    const h2 lo = {(_Float16)((x - (float)hi[0]) * LOSCALE), (_Float16)((y - (float)hi[1]) * LOSCALE)};
                    const int row = 4 * rg + (odd ? 2 : 0) + u, k = 32 * wave + 16 * t + (c & ~1);
                    *reinterpret_cast<h2*>(&sH[0][row * LDH + k]) = hi;
                    *reinterpret_cast<h2*>(&sH[1][row * LDH + k]) = lo;
                }
            }
            __syncthreads();
            f32x4 accx[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                const h8 ah = *reinterpret_cast<const h8*>(&sH[0][c * LDH + 32 * kb + 8 * rg]);
                const h8 al = *reinterpret_cast<const h8*>(&sH[1][c * LDH + 32 * kb + 8 * rg]);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    h8 bh, bl;
                    uint32_t* bhp = reinterpret_cast<uint32_t*>(&bh);
                    uint32_t* blp = reinterpret_cast<uint32_t*>(&bl);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { bhp[r] = wp[((kb * 2 + t) * 2 + 0) * 4 + r]; blp[r] = wp[((kb * 2 + t) * 2 + 1) * 4 + r]; }
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[t], 0, 0, 0);
#ifdef ONEACC
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[t], 0, 0, 0);
#else
                    accx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, accx[t], 0, 0, 0);
                    accx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, accx[t], 0, 0, 0);
#endif
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t][j] = (acc[t][j] + accx[t][j] * (1.f / LOSCALE)) * (1.f / (WSCALE * ASCALE));
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int k = 32 * wave + 16 * t + c; sA[(4 * rg + j) * 280 + (k & 3) * 68 + (k >> 2)] = h[t][j]; }
            __syncthreads();
            const float* base = sA + c * 280 + rg * 68;
#pragma unroll
            for (int q4 = 0; q4 < 16; ++q4) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(base + 4 * q4);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], w[2 * (4 * q4 + i) + t], acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) h[t][j] = elu ? __builtin_amdgcn_fmed3f(acc[t][j], __expf(acc[t][j]) - 1.f, 0.f) : acc[t][j];
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) Y[(size_t)blockIdx.x * 4096 + (4 * rg + j) * 256 + 32 * wave + 16 * t + c] = h[t][j];
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 1000;
    const int NB = 256;
    std::vector<float> W(65536), X(NB * 4096);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto& v : W) v = 0.09f * (rnd() + rnd() + rnd());           // ~ N(0, 0.09)
    for (auto& v : X) { float z = 1.5f * (rnd() + rnd()); v = z > 0 ? z : expf(z) - 1.f; if (rand() % 37 == 0) v *= 1e-4f; }
    float *dW, *dX, *dY; uint32_t* dP;
    hipMalloc(&dW, 65536 * 4); hipMalloc(&dP, 65536 * 4); hipMalloc(&dX, X.size() * 4); hipMalloc(&dY, X.size() * 4);
    hipMemcpy(dW, W.data(), 65536 * 4, hipMemcpyHostToDevice); hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_pack, dim3(256), dim3(256), 0, 0, dW, dP);
    // numerics: one step, no ELU, block 0 vs fp64
    std::vector<double> ref(4096);
    for (int r = 0; r < 16; ++r) for (int n = 0; n < 256; ++n) { double s = 0; for (int k = 0; k < 256; ++k) s += (double)X[r * 256 + k] * W[k * 256 + n]; ref[r * 256 + n] = s; }
    std::vector<float> Y(X.size());
    for (int split = 0; split < 2; ++split) {
        if (split) hipLaunchKernelGGL(k_step<true>, dim3(NB), dim3(512), 0, 0, dW, dP, dX, dY, 1, 0);
        else hipLaunchKernelGGL(k_step<false>, dim3(NB), dim3(512), 0, 0, dW, dP, dX, dY, 1, 0);
        hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost);
        double num = 0, den = 0, mx = 0, sabs = 0;
        for (int i = 0; i < 4096; ++i) { num += (Y[i] - ref[i]) * (Y[i] - ref[i]); den += ref[i] * ref[i]; mx = fmax(mx, fabs(Y[i] - ref[i])); sabs = fmax(sabs, fabs(ref[i])); }
        printf("%s: rel-L2 vs fp64 %.3e   max abs err %.3e (max |ref| %.3f)\n", split ? "fp16x2-split (3 MFMA)" : "fp32 MFMA            ", sqrt(num / den), mx, sabs);
    }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int split = 0; split < 2; ++split)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            if (split) hipLaunchKernelGGL(k_step<true>, dim3(NB), dim3(512), 0, 0, dW, dP, dX, dY, steps, 1);
            else hipLaunchKernelGGL(k_step<false>, dim3(NB), dim3(512), 0, 0, dW, dP, dX, dY, steps, 1);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("%s steps=%d: %.3f us/step\n", split ? "split" : "fp32 ", steps, 1e3 * ms / steps);
        }
    return 0;
}
