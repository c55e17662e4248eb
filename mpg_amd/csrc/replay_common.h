// Shared by the replay ring (replay_kernels.hip) and the prioritized sampler (per_kernels.hip): the ring's arrays and the gather
// of one transition (ReplayBuffer._encode_sample, buffer.py:57-68).
#pragma once
#include "mpg_common.h"

namespace {

struct Ring {
    float *obs, *act, *rew, *obs2;
    uint8_t* done;
};

// one transition: every load is issued before the first store (written as dependent load/store pairs the 15 random
// reads of a row serialise on HBM latency: 11 us for 4096 rows instead of 3)
constexpr int MAXOD = 16, MAXAD = 2;       // 16: PathTracking observations with look-ahead entries (6 + num_future_data <= 16)
template <int WOD>
__device__ __forceinline__ void gather_row_w(const Ring& r, long s, long i, int od, int ad, float* __restrict__ o_obs,
                                           float* __restrict__ o_act, float* __restrict__ o_rew,
                                           float* __restrict__ o_obs2, float* __restrict__ o_done) {
    float vo[WOD], vo2[WOD], va[MAXAD];
#pragma unroll
    for (int k = 0; k < WOD; ++k) {
        vo[k] = k < od ? r.obs[s * od + k] : 0.f;
        vo2[k] = k < od ? r.obs2[s * od + k] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < MAXAD; ++k) va[k] = k < ad ? r.act[s * ad + k] : 0.f;
    const float rew = r.rew[s];
    const uint8_t dn = r.done[s];
#pragma unroll
    for (int k = 0; k < WOD; ++k)
        if (k < od) {
            o_obs[i * od + k] = vo[k];
            o_obs2[i * od + k] = vo2[k];
        }
#pragma unroll
    for (int k = 0; k < MAXAD; ++k)
        if (k < ad) o_act[i * ad + k] = va[k];
    o_rew[i] = rew;
    if (o_done) o_done[i] = (float)dn;                     // learners cast dones to float32 (mpg_learner.py:71)
}
__device__ __forceinline__ void gather_row(const Ring& r, long s, long i, int od, int ad, float* __restrict__ o_obs,
                                           float* __restrict__ o_act, float* __restrict__ o_rew,
                                           float* __restrict__ o_obs2, float* __restrict__ o_done) {
    if (od <= 8) gather_row_w<8>(r, s, i, od, ad, o_obs, o_act, o_rew, o_obs2, o_done);      // (uniform branch)
    else gather_row_w<16>(r, s, i, od, ad, o_obs, o_act, o_rew, o_obs2, o_done);
}

}  // namespace
