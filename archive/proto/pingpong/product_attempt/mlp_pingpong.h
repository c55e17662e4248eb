// Ping-pong (two-team) kernels of the MLP engine for large batches: see mlp_pingpong.hip.  Internal to libmpg_hip.so.
#pragma once
#include <algorithm>

#include "mlp_launch.h"

namespace mlp {

struct PpFwdArgs {          // the arguments of k_forward (mlp_kernels.hip); `pack` (the packed forward image) is required
    const float* params;
    int in_dim, out_dim, rows;
    XSpec x;
    int out_tanh;
    float out_scale, sigma;
    uint32_t k0, k1, c1, c2;
    float* y;
    int ldy;
    float *h1, *h2;
    const float* pack;
    int* status;
};

// a workgroup needs at least this many row groups for the two-team schedule to pay (its prologue moves 112 KB of lo halves into
// LDS on top of the register image): 65 536 rows = 16 groups per workgroup
#ifndef MPG_PP_MIN_GROUPS_PER_WG
#define MPG_PP_MIN_GROUPS_PER_WG 6
#endif

bool pingpong_forward_available(int in_dim, int ou);
int launch_forward_pp(const PpFwdArgs& a, int ou, hipStream_t s);

}  // namespace mlp
