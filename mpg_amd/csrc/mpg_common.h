// Shared host/device helpers of libmpg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mpg_hip.h"

#define MPG_ABI_VERSION 10  // 3: caller-owned handles (mpg_wcache_t, mpg_prof_t) instead of process-wide state
                            // 4: mpg_replay_draw_t gained the pre-gathered window, mpg_env_step_store_reset_draw
                            // 5: status words (mpg_cfg_t.status, mpg_wcache_t.status), step entry points for TD3 / NADP
                            // 6: mpg_cfg_t.obs_scale has 16 entries (observations with look-ahead entries: obs_dim up to 14)
                            // 8: mpg_worker_step
                            // 9: mpg_sum_slots_strided (two-shot exchange), mpg_cfg_t.grad_opts (critics_ready_event)
                            // 10: mpg_sum_slots_sq, mpg_train_ctx_t.clip_partials_ready
                            // 7: MPG_PROF_SLOTS 10 (gradient exchange, k_clip_adam_polyak), mpg_prof_region_begin / _end

void mpg_set_error(const char* fmt, ...);

#define MPG_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            mpg_set_error(__VA_ARGS__);        \
            return MPG_EINVAL;                 \
        }                                      \
    } while (0)

// Kernel launches never synchronise; a launch-configuration error is the only thing visible here.
#define MPG_CHECK_LAUNCH(name)                                              \
    do {                                                                    \
        hipError_t e_ = hipGetLastError();                                  \
        if (e_ != hipSuccess) {                                             \
            mpg_set_error("%s: %s", name, hipGetErrorString(e_));           \
            return -(int)e_;                                                \
        }                                                                   \
    } while (0)

// optional HIP-event timing of a launch (no-ops for a null timer); slots: see include/mpg_hip.h
void mpg_prof_begin(mpg_prof_t* p, int slot, hipStream_t s);
void mpg_prof_end(mpg_prof_t* p, int slot, hipStream_t s);
inline mpg_prof_t* mpg_prof_of(const mpg_cfg_t* cfg) { return cfg ? cfg->prof : nullptr; }
inline int* mpg_status_of(const mpg_cfg_t* cfg) { return cfg ? cfg->status : nullptr; }

static inline hipStream_t mpg_stream(mpg_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// ---- Philox4x32-10 (Salmon et al. 2011), counter-based: (key, counter) -> 4 x u32 -------------------
struct Philox4 {
    uint32_t v[4];
};
// word k (0..3) of a draw by selects.  NEVER index v[] with a run-time value in device code: the compiler moves the four
// words to LDS, addresses them by the flattened thread id, and reads the workgroup size for that from the dispatch packet
// - which lives in HOST memory.  That one scalar load cost the lanes that draw the minibatch 7 to 30 us per launch
// (round 2; tools/kernel_resources.py --dispatch-ptr lists kernels that read the packet).
__host__ __device__ static inline uint32_t philox_word(const Philox4& p, int k) {
    const uint32_t lo = (k & 1) ? p.v[1] : p.v[0], hi = (k & 1) ? p.v[3] : p.v[2];
    return (k & 2) ? hi : lo;
}

__host__ __device__ static inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                        uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    Philox4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

// u32 -> float in the open interval (0,1): 24 random mantissa bits, centred.
__host__ __device__ static inline float u01(uint32_t u) { return ((float)(u >> 8) + 0.5f) * 5.9604644775390625e-08f; }
#ifdef __HIPCC__
// y + sigma * N(0, 1), the normal deviate by Box-Muller from two uniforms on the HARDWARE transcendentals (v_log_f32 = log2,
// v_sqrt_f32, v_cos_f32 whose argument is in revolutions): single instructions with explicitly separate roundings, so that every
// translation unit produces the same bits whatever its -ffp-contract mode.  (The library's logf / cosf are expanded by the backend
// with or without fused steps depending on that mode - the worker's policy pass exists in two translation units, mlp_kernels.hip
// and env_path_tracking.hip, which must agree bit for bit.)
__device__ __forceinline__ float add_gauss_noise(float y, float sigma, float u1, float u2) {
    const float r2 = __builtin_amdgcn_logf(u1) * -1.3862943611198906f;       // -2 ln u1 = (-2 ln 2) log2 u1  (u1 in (0, 1))
    const float n = __builtin_amdgcn_sqrtf(r2) * __builtin_amdgcn_cosf(u2);   // sqrt(-2 ln u1) cos(2 pi u2)
    return __builtin_fmaf(sigma, n, y);
}
#endif

// sum over the 256 threads of a block of one value each, fixed tree order; `red` = 256 floats of LDS.  Shared by every
// kernel that produces the clip's per-block partial sums of squares, so that they are bit-identical.
#ifdef __HIPCC__
__device__ __forceinline__ float mpg_block_sum256(float a, float* red) {
    red[threadIdx.x] = a;
    __syncthreads();
#pragma unroll
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    return red[0];
}
#endif
