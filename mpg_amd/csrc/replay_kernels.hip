// K9 (part 1): device-resident replay ring - add, uniform index draw, gather.
// Reference: buffer.py:21-91 (ReplayBuffer: python-list ring, random.randint sampling with replacement,
// _encode_sample).  Transitions are stored SoA-of-rows: obs [cap][obs_dim], act [cap][act_dim], rew [cap],
// obs2 [cap][obs_dim], done [cap] (uint8), RAW rewards and observations like the reference.
#include "replay_common.h"

namespace {

__global__ void k_add(int capacity, int next_idx, int n, int od, int ad, const float* __restrict__ s_obs,
                      const float* __restrict__ s_act, const float* __restrict__ s_rew,
                      const float* __restrict__ s_obs2, const uint8_t* __restrict__ s_done, Ring r) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long d = (long)((next_idx + i) % capacity);      // buffer.py:49-55
    for (int k = 0; k < od; ++k) {
        r.obs[d * od + k] = s_obs[(long)i * od + k];
        r.obs2[d * od + k] = s_obs2[(long)i * od + k];
    }
    for (int k = 0; k < ad; ++k) r.act[d * ad + k] = s_act[(long)i * ad + k];
    r.rew[d] = s_rew[i];
    r.done[d] = s_done ? s_done[i] : 1;
}

__global__ void k_gather(int n, const int* __restrict__ idx, int od, int ad, Ring r, float* __restrict__ o_obs,
                         float* __restrict__ o_act, float* __restrict__ o_rew, float* __restrict__ o_obs2,
                         float* __restrict__ o_done) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long s = idx[i];                                  // buffer.py:57-68
    gather_row(r, s, i, od, ad, o_obs, o_act, o_rew, o_obs2, o_done);
}

__global__ void k_uniform_idx(int n_storage, int n, uint32_t k0, uint32_t k1, uint32_t c1, uint32_t c2,
                              int* __restrict__ idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Philox4 p = philox4x32_10((uint32_t)(i >> 2), c1, c2, 0x1d5u, k0, k1);
    // random.randint(0, len-1), buffer.py:70-71: multiply-shift maps a 32-bit draw to [0, n_storage)
    idx[i] = (int)(((uint64_t)philox_word(p, i & 3) * (uint64_t)n_storage) >> 32);
}

// ReplayBuffer.sample (buffer.py:70-78): index draw + gather in one launch (same Philox stream as k_uniform_idx)
__global__ void k_sample_gather(int n_storage, int n, uint32_t k0, uint32_t k1, uint32_t c1, uint32_t c2, int od, int ad, Ring r,
                                int* __restrict__ idx, float* __restrict__ o_obs, float* __restrict__ o_act,
                                float* __restrict__ o_rew, float* __restrict__ o_obs2, float* __restrict__ o_done) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Philox4 p = philox4x32_10((uint32_t)(i >> 2), c1, c2, 0x1d5u, k0, k1);
    const long s = (long)(((uint64_t)philox_word(p, i & 3) * (uint64_t)n_storage) >> 32);
    idx[i] = (int)s;
    gather_row(r, s, i, od, ad, o_obs, o_act, o_rew, o_obs2, o_done);
}

}  // namespace

extern "C" int mpg_replay_add(int capacity, int next_idx, int n, int obs_dim, int act_dim, const float* s_obs,
                              const float* s_act, const float* s_rew, const float* s_obs2, const uint8_t* s_done,
                              float* obs, float* act, float* rew, float* obs2, uint8_t* done, mpg_stream_t stream) {
    MPG_REQUIRE(capacity > 0 && n > 0 && n <= capacity && next_idx >= 0 && next_idx < capacity && s_obs && s_act &&
                    s_rew && s_obs2 && obs && act && rew && obs2 && done,
                "mpg_replay_add: bad argument");
    Ring r{obs, act, rew, obs2, done};
    hipLaunchKernelGGL(k_add, dim3((n + 255) / 256), dim3(256), 0, mpg_stream(stream), capacity, next_idx, n, obs_dim,
                       act_dim, s_obs, s_act, s_rew, s_obs2, s_done, r);
    MPG_CHECK_LAUNCH("k_add");
    return MPG_OK;
}

extern "C" int mpg_replay_gather(int n, const int* idx, int obs_dim, int act_dim, const float* obs, const float* act,
                                 const float* rew, const float* obs2, const uint8_t* done, float* o_obs, float* o_act,
                                 float* o_rew, float* o_obs2, float* o_done, mpg_stream_t stream) {
    MPG_REQUIRE(n > 0 && idx && obs && act && rew && obs2 && done && o_obs && o_act && o_rew && o_obs2,
                "mpg_replay_gather: bad argument");
    MPG_REQUIRE(obs_dim >= 1 && obs_dim <= MAXOD && act_dim >= 1 && act_dim <= MAXAD, "mpg_replay_gather: obs_dim <= 16, act_dim <= 2");
    Ring r{const_cast<float*>(obs), const_cast<float*>(act), const_cast<float*>(rew), const_cast<float*>(obs2),
           const_cast<uint8_t*>(done)};
    hipLaunchKernelGGL(k_gather, dim3((n + 255) / 256), dim3(256), 0, mpg_stream(stream), n, idx, obs_dim, act_dim, r,
                       o_obs, o_act, o_rew, o_obs2, o_done);
    MPG_CHECK_LAUNCH("k_gather");
    return MPG_OK;
}

extern "C" int mpg_uniform_indices(int n_storage, int n, uint64_t seed, uint64_t ctr, int* idx, mpg_stream_t stream) {
    MPG_REQUIRE(n_storage > 0 && n > 0 && idx, "mpg_uniform_indices: bad argument");
    hipLaunchKernelGGL(k_uniform_idx, dim3((n + 255) / 256), dim3(256), 0, mpg_stream(stream), n_storage, n,
                       (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)ctr, (uint32_t)(ctr >> 32), idx);
    MPG_CHECK_LAUNCH("k_uniform_idx");
    return MPG_OK;
}

extern "C" int mpg_replay_sample_uniform(int n_storage, int n, uint64_t seed, uint64_t ctr, int obs_dim, int act_dim,
                                         const float* obs, const float* act, const float* rew, const float* obs2,
                                         const uint8_t* done, int* idx, float* o_obs, float* o_act, float* o_rew, float* o_obs2,
                                         float* o_done, mpg_stream_t stream) {
    MPG_REQUIRE(n_storage > 0 && n > 0 && obs && act && rew && obs2 && done && idx && o_obs && o_act && o_rew && o_obs2,
                "mpg_replay_sample_uniform: bad argument");
    MPG_REQUIRE(obs_dim >= 1 && obs_dim <= MAXOD && act_dim >= 1 && act_dim <= MAXAD, "mpg_replay_sample_uniform: obs_dim <= 16, act_dim <= 2");
    Ring r{const_cast<float*>(obs), const_cast<float*>(act), const_cast<float*>(rew), const_cast<float*>(obs2),
           const_cast<uint8_t*>(done)};
    hipLaunchKernelGGL(k_sample_gather, dim3((n + 255) / 256), dim3(256), 0, mpg_stream(stream), n_storage, n, (uint32_t)seed,
                       (uint32_t)(seed >> 32), (uint32_t)ctr, (uint32_t)(ctr >> 32), obs_dim, act_dim, r, idx, o_obs, o_act, o_rew,
                       o_obs2, o_done);
    MPG_CHECK_LAUNCH("k_sample_gather");
    return MPG_OK;
}
