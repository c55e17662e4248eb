// Internal interface between the env entry points (env_path_tracking.hip) and the second real environment.
#pragma once
#include "mpg_common.h"

namespace cart_pole {   // env_cart_pole.hip: analytic RK4 statement of inverted_pendulum_conti.xml
int reset_from_obs(int n, int obs_dim, float* state, const float* init_obs, hipStream_t s);
int reset(int n, int obs_dim, float* state, const uint8_t* done_mask, uint64_t seed, uint64_t ctr, float* obs, hipStream_t s);
int step(int n, int obs_dim, float* state, const float* action, float* obs, float* reward, uint8_t* done, uint8_t* done_intended,
         hipStream_t s);
int step_store_reset(int n, int obs_dim, float* state, const float* action, int capacity, int next_idx, float* ring_obs,
                     float* ring_act, float* ring_rew, float* ring_obs2, uint8_t* ring_done, uint64_t seed, uint64_t ctr,
                     float* obs_out, uint8_t* done_out, hipStream_t s);
}  // namespace cart_pole
