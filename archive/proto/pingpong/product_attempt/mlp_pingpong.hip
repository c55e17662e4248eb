// PING-PONG geometry of the 2x256 ELU MLP engine for LARGE batches (round 4): the chain-free network kernels (forward, input-side
// backward) when a workgroup has several row groups to take - TD3 at B = 65 536, model.py:39-43 / policy.py:193-241 evaluated by
// learners/td3.py:69-134.
//
// The lock-step engine of mlp_core.h puts all eight waves of a workgroup through the same phase at the same time: layer 1 + ELU +
// fp16 split + image store (vector), barrier, the 256 x 256 matrix block (matrix pipe), epilogue (vector), barrier.  A SIMD's
// matrix pipe idles during the vector phases and its vector issue during the matrix block: a row group costs 4.3 - 4.5 k cycles for
// 1.8 k cycles of matrix pipe in every lock-step structure tried.  Here a workgroup is two TEAMS of four waves (team =
// wave >> 2: the two waves of a SIMD are w and w + 4, so every SIMD hosts one wave of each team).  A team evaluates row groups on
// its own, half a period behind the other team: while one issues its matrix block (96 MFMAs per wave = 1536 cycles of its SIMD's
// matrix pipe), the other does the vector work around it (epilogue of its previous group, layer 1 of its next) - matrix pipe and
// vector issue of every SIMD busy at the same time with DIFFERENT waves.  What makes that fit:
//   * a team has four waves, so wave q owns 64 hidden columns (4 tiles of 16): the fp16 HI halves of its 256 x 64 slice of W2
//     are register-stationary (128 VGPRs), the LO halves are read from a 112 KB LDS image shared by both teams (k-blocks 0..6;
//     k-block 7 stays in 16 VGPRs: the whole lo image does not fit beside the activation images) - the register budget is the
//     lock-step engine's, the weight image is held ONCE per CU instead of once per team;
//   * the activation image is the TRANSPOSED one (mlp_core.h): ds_write_b64 per tile and image, ds_read_b64_tr_b16 reads - the
//     vector phase runs at half rate beside the partner's MFMAs (a 16x16x32 MFMA holds the SIMD's vector issue for 8 of its 16
//     cycles, MI355X_MICROARCH.md), so every vector instruction removed from it counts twice;
//   * TEAM-LOCAL barriers (an LDS arrival counter per team, polled with s_sleep): a team never waits for the other one's phase.
// Both images come from the caller's packed weight cache in its existing lane order (a ping-pong wave's tile t of k-block kb is
// fragment (kb, t & 1) of lock-step wave 2 q + (t >> 1)): no second packed layout, Adam keeps updating one image per direction.
// Measured (archive/proto/pingpong/, the stand-alone forward pass at 65 536 rows): 30.3 - 31.5 us against 37.0 - 38.5 us for the
// lock-step pairs; the matrix pipe goes from ~50 % to ~70 % busy.  Used by launch_forward / launch_backward once a workgroup has at
// least MPG_PP_MIN_GROUPS_PER_WG row groups; results agree with the lock-step kernels to float32 rounding (the output layer's
// partial sums are grouped by four waves instead of eight) and the G16 stashes are bit-identical.
#include "mlp_pingpong.h"

namespace mlp {

#ifdef MPG_SPLIT

namespace {

constexpr int TW = 4;                 // waves per team
constexpr int NT = 4;                 // 16-column tiles per wave
constexpr int LO_KB = 7;              // k-blocks of the lo image kept in LDS (the last one stays in registers)
constexpr int LO_FLOATS = TW * LO_KB * NT * 256;      // 28 672 floats = 112 KB

struct PpLane {
    int tid, lane, wave, team, q, c, rg;
    __device__ PpLane() {
        tid = threadIdx.x; lane = tid & 63; wave = tid >> 6; team = wave >> 2; q = wave & 3; c = lane & 15; rg = lane >> 4;
    }
};

// f32x4 index of fragment (k-block kb, tile t of ping-pong wave q, part 0 = hi / 1 = lo) in the lock-step packed image
__device__ __forceinline__ int pp_frag(const PpLane& P, int kb, int t, int part) {
    return ((2 * P.q + (t >> 1)) * 32 + (kb * 2 + (t & 1)) * 2 + part) * 64 + P.lane;
}

// the wave's stationary pieces: hi halves (128 VGPRs), lo halves of k-block 7 (16 VGPRs), its share of the LDS lo image
__device__ __forceinline__ void pp_load_weights(const float* __restrict__ pack, const PpLane& P, float (&whi)[128], float (&wlo7)[16],
                                                float* sLo) {
    const f32x4* pk = reinterpret_cast<const f32x4*>(pack);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4 h = pk[pp_frag(P, kb, t, 0)];
#pragma unroll
            for (int e = 0; e < 4; ++e) whi[(kb * NT + t) * 4 + e] = h[e];
        }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const f32x4 l = pk[pp_frag(P, 7, t, 1)];
#pragma unroll
        for (int e = 0; e < 4; ++e) wlo7[t * 4 + e] = l[e];
    }
    for (int v = P.team; v < LO_KB * NT; v += 2)          // wave (team, q) copies half of slice q's 28 lo fragments
        *reinterpret_cast<f32x4*>(sLo + ((P.q * LO_KB + v / NT) * NT + v % NT) * 256 + P.lane * 4) = pk[pp_frag(P, v / NT, v % NT, 1)];
}

// the matrix block of one row group: acc[t] = (A image of the team) x (the wave's 256 x 64 slice), hi*hi + hi*lo + lo*hi per k-block
// in the order of the lock-step engine (bit-identical accumulators).  The LDS reads are software-pipelined BY HAND and fenced with
// scheduling barriers: per k-block [hi*hi x 4] [hi*lo x 4] [lo*hi x 4]; the lo weight fragments of block kb + 1 are requested right
// behind the hi*lo products of block kb into the registers those have just freed and are first used 8 MFMAs (128 cycles) later;
// the lo activation fragment likewise behind the lo*hi products; the hi activation fragment alternates between two registers.
// (Left to the scheduler at this register pressure, every lo fragment went through ONE register quadruple with its LDS latency
// exposed in front of its MFMA: 28 stalls per block - the product kernel measured 6 us slower than its prototype for that.)
__device__ __forceinline__ void pp_matrix_block(const char* img, const float* sLo, const PpLane& P, const float (&whi)[128],
                                                const float (&wlo7)[16], f32x4 (&acc)[NT]) {
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    // lane 4 q' + p' of its 16-lane group supplies block row q' (k = 32 kb + 8 rg + q', + 4 for the second read), chunk p'
    const int s0 = 16 * (P.rg >> 1) + 4 * (P.rg & 1) + ((P.lane >> 2) & 3);
    const char* t0 = img + tr_byte(s0, P.lane & 3);
    const char* t1 = img + tr_byte(s0 + 8, P.lane & 3);
    const float* lo0 = sLo + (P.q * LO_KB * NT) * 256 + P.lane * 4;
    auto hfrag = [&](int kb, int t) {
        const int v = (kb * NT + t) * 4;
        return __builtin_bit_cast(f16x8, f32x4{whi[v], whi[v + 1], whi[v + 2], whi[v + 3]});
    };
    auto a_frag = [&](int kb, int part) {
        const s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(t0 + part * TR_IMG_BYTES + 1024 * kb));
        const s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(t1 + part * TR_IMG_BYTES + 1024 * kb));
        return __builtin_bit_cast(f16x8, __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto lo_frag = [&](int kb, int t) {
        return __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(lo0 + (kb * NT + t) * 256));
    };
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 ah = a_frag(0, 0), al = a_frag(0, 1), ah_next = ah, wl[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wl[t] = lo_frag(0, t);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        if (kb + 1 < 8) ah_next = a_frag(kb + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, hfrag(kb, t), acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl[t], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 1 < LO_KB) {
#pragma unroll
            for (int t = 0; t < NT; ++t) wl[t] = lo_frag(kb + 1, t);
        } else if (kb + 1 == LO_KB) {
#pragma unroll
            for (int t = 0; t < NT; ++t) wl[t] = __builtin_bit_cast(f16x8, f32x4{wlo7[t * 4], wlo7[t * 4 + 1], wlo7[t * 4 + 2], wlo7[t * 4 + 3]});
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, hfrag(kb, t), acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 1 < 8) al = a_frag(kb + 1, 1);
        ah = ah_next;
    }
}

// C-layout values v[t][j] (column 64 q + 16 t + c, rows 4 rg + j), times A_SCALE, into the team's transposed image (hi, then lo)
__device__ __forceinline__ void pp_store_image(char* img, const PpLane& P, const float (&v)[NT][4]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        unsigned h01, l01, h23, l23;
        split2_mix(v[t][0], v[t][1], A_SCALE, h01, l01);
        split2_mix(v[t][2], v[t][3], A_SCALE, h23, l23);
        const int byte = tr_byte(tr_slot(64 * P.q + 16 * t + P.c), P.rg);
        *reinterpret_cast<u32x2*>(img + byte) = u32x2{h01, h23};
        *reinterpret_cast<u32x2*>(img + TR_IMG_BYTES + byte) = u32x2{l01, l23};
    }
}

// team-local barrier: every wave of the team adds to the team's LDS counter and polls it (s_sleep between polls).  The LDS
// serves a wave's operations in order, so a wave that sees the counter at its target also sees the stores each arriving wave
// issued in front of its add.
struct TeamBarrier {
    unsigned* cnt;
    unsigned target;
    __device__ TeamBarrier(unsigned* c) : cnt(c), target(0) {}
    __device__ __forceinline__ void wait(int lane) {
        target += TW;
        __builtin_amdgcn_s_waitcnt(0xC07F);                                 // lgkmcnt(0): this wave's LDS stores are in
        if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
};

// ---------------------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------------------
template <int IN, int OU>
__global__ void __launch_bounds__(NTHREAD, 2) k_forward_pp(const PpFwdArgs a) {
    static_assert(IN <= 8, "the ping-pong forward takes first layers up to 8 wide (two fp32 MFMA k-steps)");
    __shared__ __attribute__((aligned(16))) float sLo[LO_FLOATS];
    __shared__ __attribute__((aligned(16))) char sImg[2][2 * TR_IMG_BYTES];
    __shared__ float sPart[2][2][TW * GROUP * MAXOUT];          // [team][parity]
    __shared__ unsigned sTeamCnt[2];
    const PpLane P;
    const Net net = make_net(a.params, a.in_dim, a.out_dim);
    if (P.tid < 2) sTeamCnt[P.tid] = 0;
    float whi[128], wlo7[16];
    pp_load_weights(a.pack, P, whi, wlo7, sLo);
    float w1p[2][NT], b1[NT], b2[NT], w3[NT][OU];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = 64 * P.q + 16 * t + P.c;
#pragma unroll
        for (int s = 0; s < 2; ++s) w1p[s][t] = (4 * s + P.rg) < a.in_dim ? net.W1[(4 * s + P.rg) * H + col] : 0.f;
        b1[t] = net.b1[col];
        b2[t] = net.b2[col];
#pragma unroll
        for (int o = 0; o < OU; ++o) w3[t][o] = net.W3[col * a.out_dim + o];
    }
    const float b3v = net.b3[P.lane % OU];
    char* img = sImg[P.team];
    const long ngroups = (a.rows + GROUP - 1) / GROUP;
    // team T of workgroup b takes groups 2 b + T, 2 b + T + 2 gridDim.x, ...; every team runs the same number of iterations
    const long gstep = 2L * gridDim.x, g_first = 2L * blockIdx.x + P.team;
    const long n_it = (ngroups + gstep - 1) / gstep;
    // this lane's layer-1 A operand of group g: x[row c][column 4 s + rg] (zero beyond the batch / the input width).  Which array
    // a column comes from, its stride and its scale depend on the lane only: settled once; the loads issued under a matrix block
    // are RAW (nothing is computed from them there - a multiply would drag their memory latency in front of the matrix block)
    const float* xb[2];
    long xld[2];
    float xsc[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int i = 4 * s + P.rg;
        const bool from0 = i < a.x.d0;
        xb[s] = i < a.in_dim ? (from0 ? a.x.x0 + i : a.x.x1 + (i - a.x.d0)) : nullptr;
        xld[s] = from0 ? a.x.ld0 : a.x.ld1;
        float sc = 1.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) sc = (from0 && i == k) ? a.x.scale[k] : sc;      // (constant indices: no kernel-argument load)
        xsc[s] = sc;
    }
    auto x_load = [&](long g, float (&xr)[2]) {
        const long gr = g * GROUP + P.c;
#pragma unroll
#ifdef MPG_PP_AB_SIMPLEX
        for (int s = 0; s < 2; ++s) xr[s] = (g < ngroups && gr < a.rows && 4 * s + P.rg < a.in_dim) ? a.x.x0[gr * a.in_dim + 4 * s + P.rg] : 0.f;
#else
        for (int s = 0; s < 2; ++s) xr[s] = (xb[s] && g < ngroups && gr < a.rows) ? xb[s][gr * xld[s]] : 0.f;
#endif
    };
    f32x4 acc[NT];
    float xa[2];
    float zmax = 0.f;
    bool saw_nan = false;
    // a NaN (or an infinity) among a row's inputs must stay a NaN in its outputs although the ELU's v_med3 drops it (row_poison,
    // mlp_core.h): the row's layer-1 pre-activation is then a NaN in every column, so 0 * z of the lane's own four rows, carried to
    // the group's epilogue and added to its output partials, does it with no cross-lane traffic
    float pz[4] = {0.f, 0.f, 0.f, 0.f};
    x_load(g_first, xa);
    __syncthreads();                                          // the lo image and the team counters are set
    TeamBarrier bar(&sTeamCnt[P.team]);
    // Iteration i of a team, group g = g_first + i gstep: [vector interval: epilogue of i - 1, output of i - 2, layer 1 of i]
    // team barrier [matrix interval: the matrix block of i, next inputs requested] team barrier.  Team 1 starts half a period late
    // (one idle poll of its own counter costs nothing: the teams drift apart by themselves as soon as they share the matrix pipe).
    for (long i = 0; i <= n_it; ++i) {
        const long g = g_first + i * gstep;
        const int par = (int)((i + 1) & 1);
        if (i > 0) {                                          // ---- epilogue of iteration i - 1 ----
            const long gp = g - gstep;
            float h2[NT][4];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float z = fmaf(acc[t][j], 1.f / (W_SCALE * A_SCALE), b2[t]);
                    h2[t][j] = __builtin_amdgcn_fmed3f(z, __builtin_amdgcn_exp2f(z * 1.4426950408889634f) - 1.f, 0.f);
                }
#ifndef MPG_PP_AB_NOSTASH
                if (a.h2 && gp < ngroups)
#else
                if (false)
#endif
                    reinterpret_cast<f32x4*>(a.h2)[(gp * 16 + 4 * P.q + t) * 64 + P.lane] = f32x4{h2[t][0], h2[t][1], h2[t][2], h2[t][3]};
            }
            float p[OU][4];
#pragma unroll
            for (int o = 0; o < OU; ++o)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    p[o][j] = row_allreduce16(fmaf(h2[1][j], w3[1][o], h2[0][j] * w3[0][o]) + fmaf(h2[3][j], w3[3][o], h2[2][j] * w3[2][o]));
            if (P.c == 0) {
#pragma unroll
                for (int o = 0; o < OU; ++o)
#pragma unroll
                    for (int j = 0; j < 4; ++j) sPart[P.team][par][(P.q * GROUP + 4 * P.rg + j) * MAXOUT + o] = p[o][j] + pz[j];
            }
        }
        if (i >= 2) {                                         // ---- output of iteration i - 2 (its partials: parity i & 1) ----
            const long go = g - 2 * gstep;
            if (P.q == 0 && P.lane < GROUP * OU && go < ngroups) {
                const int row = P.lane / OU, o = P.lane % OU;
                const long gr = go * GROUP + row;
                if (gr < a.rows) {
                    float z = b3v;
#pragma unroll
                    for (int w = 0; w < TW; ++w) z += sPart[P.team][(int)(i & 1)][(w * GROUP + row) * MAXOUT + o];
#ifdef MPG_PP_AB_PLAINOUT
                    float y = z;
                    if (false) {
#else
                    float y = a.out_tanh ? a.out_scale * tanhf(z) : z;
                    if (a.sigma > 0.f) {   // OffPolicyWorker.sample: action += N(0, sigma), worker.py:97-98 (k_forward's stream)
#endif
                        Philox4 ph = philox4x32_10((uint32_t)gr, a.c1, a.c2, 0x5eedu + (uint32_t)o, a.k0, a.k1);
                        y += a.sigma * sqrtf(-2.f * logf(u01(ph.v[0]))) * cosf(6.283185307179586f * u01(ph.v[1]));
                    }
                    saw_nan |= y != y;
                    a.y[gr * a.ldy + o] = y;
                }
            }
        }
        if (i < n_it) {                                       // ---- layer 1 of iteration i ----
            xa[0] *= xsc[0];
            xa[1] *= xsc[1];
            saw_nan |= (xa[0] != xa[0]) | (xa[1] != xa[1]);
            f32x4 z[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) z[t] = f32x4{b1[t], b1[t], b1[t], b1[t]};
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < NT; ++t) z[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], w1p[s][t], z[t], 0, 0, 0);
#ifndef MPG_PP_AB_NOCHECKS
#pragma unroll
            for (int j = 0; j < 4; ++j) pz[j] = z[0][j] * 0.f;
#endif
            float h1[NT][4];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    h1[t][j] = __builtin_amdgcn_fmed3f(z[t][j], __builtin_amdgcn_exp2f(z[t][j] * 1.4426950408889634f) - 1.f, 0.f);
#ifndef MPG_PP_AB_NOCHECKS
                    zmax = fmaxf(zmax, h1[t][j]);
#endif
                }
#ifndef MPG_PP_AB_NOSTASH
                if (a.h1 && g < ngroups)
#else
                if (false)
#endif
                    reinterpret_cast<f32x4*>(a.h1)[(g * 16 + 4 * P.q + t) * 64 + P.lane] = f32x4{h1[t][0], h1[t][1], h1[t][2], h1[t][3]};
            }
            pp_store_image(img, P, h1);
        }
        bar.wait(P.lane);
        if (i == n_it) break;
        x_load(g + gstep, xa);                                // the next group's inputs travel under the matrix block
        pp_matrix_block(img, sLo, P, whi, wlo7, acc);
        bar.wait(P.lane);
    }
    // the last output: iteration n_it - 1, partials written in the last vector interval (parity (n_it + 1) & 1), poison in pz_mid
    {
        const long go = g_first + (n_it - 1) * gstep;
        if (n_it >= 1 && P.q == 0 && P.lane < GROUP * OU && go < ngroups) {
            const int row = P.lane / OU, o = P.lane % OU;
            const long gr = go * GROUP + row;
            if (gr < a.rows) {
                float z = b3v;
#pragma unroll
                for (int w = 0; w < TW; ++w) z += sPart[P.team][(int)((n_it + 1) & 1)][(w * GROUP + row) * MAXOUT + o];
#ifdef MPG_PP_AB_PLAINOUT
                float y = z;
                if (false) {
#else
                float y = a.out_tanh ? a.out_scale * tanhf(z) : z;
                if (a.sigma > 0.f) {
#endif
                    Philox4 ph = philox4x32_10((uint32_t)gr, a.c1, a.c2, 0x5eedu + (uint32_t)o, a.k0, a.k1);
                    y += a.sigma * sqrtf(-2.f * logf(u01(ph.v[0]))) * cosf(6.283185307179586f * u01(ph.v[1]));
                }
                saw_nan |= y != y;
                a.y[gr * a.ldy + o] = y;
            }
        }
    }
    report_activation_range(a.status, zmax);
    if (a.status && saw_nan) atomicOr(a.status, MPG_STATUS_NAN);
}

}  // namespace

bool pingpong_forward_available(int in_dim, int ou) { return (in_dim == 6 && ou == 2) || (in_dim == 8 && ou == 1) || (in_dim == 4 && ou == 1) || (in_dim == 5 && ou == 1) || (in_dim == 6 && ou == 1); }

int launch_forward_pp(const PpFwdArgs& a, int ou, hipStream_t s) {
    const long ngroups = (a.rows + GROUP - 1) / GROUP;
    const int grid = (int)std::min<long>(256, (ngroups + 1) / 2);
#define CALL(I, O) hipLaunchKernelGGL((k_forward_pp<I, O>), dim3(grid), dim3(NTHREAD), 0, s, a)
    if (a.in_dim == 6 && ou == 2) CALL(6, 2);
    else if (a.in_dim == 8 && ou == 1) CALL(8, 1);
    else if (a.in_dim == 4 && ou == 1) CALL(4, 1);
    else if (a.in_dim == 5 && ou == 1) CALL(5, 1);
    else if (a.in_dim == 6 && ou == 1) CALL(6, 1);
    else { mpg_set_error("launch_forward_pp: unsupported network shape in=%d used-out=%d", a.in_dim, ou); return MPG_EINVAL; }
#undef CALL
    MPG_CHECK_LAUNCH("k_forward_pp");
    return MPG_OK;
}

#else   // exact-fp32 engine: no ping-pong kernels (the lock-step kernels serve every size)

bool pingpong_forward_available(int, int) { return false; }
int launch_forward_pp(const PpFwdArgs&, int, hipStream_t) { mpg_set_error("launch_forward_pp: not built (MPG_F32_MFMA)"); return MPG_EINVAL; }

#endif

}  // namespace mlp
