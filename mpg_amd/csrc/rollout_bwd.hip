// Reverse sweep of the fused n-step model rollout (see rollout_kernels.hip for the overview and the references).
#include "rollout_common.h"

namespace rollout {
namespace {

// WIDE: observations with look-ahead entries (see the forward sweep): the adjoint of a MODEL observation's look-ahead entries
// folds into the adjoint of entry ENV::FUT_SRC, which they copy; the start observation's are inputs.
// THIN: the thin parameter gradients (dW1, db1, db2, dW3, db3) of EVERY step are accumulated here (rollout_common.h,
// thin_floats): used with stash_all (NADP) when M == 1.
template <class ENV, bool PK, bool WIDE = false, bool THIN = false>
__global__ void __launch_bounds__(NTHREAD, 2) k_rollout_bwd(const RollBwdArgs a) {
    constexpr int OBS = ENV::OBS, ACT = ENV::ACT;
    static_assert(!(THIN && WIDE), "THIN is built for the base observation widths");
    constexpr int NIN = WIDE ? 16 : OBS, XSW = xs_of<NIN>();
    // per-lane running sums of the thin gradients, in LDS (registers: none to spare): [quad][thread] float4, slot order
    // gb1[2] gb2[2] gW3[2][ACT] gW1[2][OBS] (+ padding); the own lanes' db3 in sB3; the step's scaled network inputs of the 16
    // rows in sXin, double-buffered by the parity of a RUNNING step count (written at the top of a step, read behind its last
    // barrier; the count runs across the groups of a workgroup: with `t & 1` and an even horizon the first step of the next
    // group (t = n) wrote the buffer the slower waves were still reading for step 0 of the previous one - ADVICE r4)
    constexpr int NACC = 4 + 2 * ACT + 2 * OBS, TQ = (NACC + 3) / 4;
    __shared__ __attribute__((aligned(16))) float sThin[THIN ? TQ * NTHREAD * 4 : 4];
    __shared__ __attribute__((aligned(16))) float sXin[THIN ? 2 * GROUP * 8 : 4];
    __shared__ float sB3[THIN ? GROUP * 2 : 1];
    const int nf = WIDE ? a.obs_dim - OBS : 0, OD = OBS + nf, QIN = OD + ACT;
    __shared__ __attribute__((aligned(16))) float smem[2 * A_IMG + GROUP * MAXOUT + NWAVE * GROUP * XSW];
    float* sA = smem;
    float* sA1 = sA + A_IMG;
    float* sD3 = sA1 + A_IMG;
    float* sPartX = sD3 + GROUP * MAXOUT;
    // carry state of the 16 trajectory lanes between steps (adjoint of the next obs, record of the next step): kept in
    // LDS because registers are allocated for all 512 lanes while only 16 use them (the kernel sits at the 256 VGPR limit)
    __shared__ __attribute__((aligned(16))) float sCarry[GROUP * 16];
    // dL/d(raw reward) per step from LDS: `a.rho[t]` with a run-time t is a scalar load from the kernel-argument segment plus a
    // wait on the serial chain of EVERY step (and a loop over `a.sel[ks]` with a run-time ks is one per slice)
    __shared__ float sRho[MAXN];
    const Lane L;
    const int tid = threadIdx.x;
    prefer_young_waves();
    if (tid < MAXN) {
        float rv = 0.f;
#pragma unroll
        for (int k = 0; k < MAXN; ++k) rv = tid == k ? a.rho[k] : rv;       // (constant indices: SGPR reads, no memory)
        sRho[tid] = rv;
    }
    unsigned selmask = 0;                      // bit t: step t is a selected slice
#pragma unroll
    for (int ks = 0; ks < MAXSEL; ++ks)
        if (ks < a.n_sel) selmask |= 1u << a.sel[ks];
    const Net net = make_net(a.policy, OD, 2 * ACT);
    float w2t[128];
    SmallRegs<NIN, ACT> r;
    if constexpr (PK) load_w2_packed(a.pack, L, w2t); else load_w2_bwd(net.W2, L, w2t);
    load_small<NIN, ACT>(net, L, r);
    const long R = (long)a.rows * a.M;
    const long ngroups = (R + GROUP - 1) / GROUP;
    if constexpr (THIN) {
#pragma unroll
        for (int q = 0; q < TQ; ++q) reinterpret_cast<f32x4*>(sThin)[q * NTHREAD + tid] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tid < GROUP) sB3[2 * tid] = sB3[2 * tid + 1] = 0.f;          // (each trajectory lane owns its two slots)
    }
#ifdef MPG_STAMP
    if ((tid & 63) == 0) {
        for (int k = 0; k < 10; ++k) g_st_acc[tid >> 6][k] = 0;
        g_st_prev[tid >> 6] = __builtin_amdgcn_s_memtime();
    }
#endif
    int xpar = 0;                              // THIN: which half of sXin this step uses (flips once per processed step)
    for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const long tr = g * GROUP + tid;
        const bool own = tid < GROUP, live = own && tr < R;
        if (tid < GROUP) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sCarry[tid * 16 + i] = 0.f;   // [0..8): dL/d(obs_{t+1}), [8..16): record t+1
        }
        float lam[8];
        // Software pipeline over the steps: the (obs | action) record and the h2 stash of step t-1 are requested while
        // step t computes, so that no HBM / L2 latency sits on the serial chain.
        float rec_cur[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float rec_pre[8];
        float h2_cur[2][4], h2_pre[2][4];
        if (live) {
            const f32x4* rp = reinterpret_cast<const f32x4*>(a.SA + ((long)a.n * R + tr) * SAW);
            const f32x4 r0 = rp[0], r1 = rp[1];
#pragma unroll
            for (int i = 0; i < 4; ++i) { rec_cur[i] = r0[i]; rec_cur[4 + i] = r1[i]; }
        }
        stash_load(a.H2, (long)a.n * ngroups + g, L, h2_cur);
        for (int t = a.n; t >= 0; --t) {
            float h1[2][4];
            // h1 of this step is requested at the TOP of the step: nothing older is pending here (the step before consumed its
            // prefetches when it copied them), nothing before the matrix block waits on the vector-memory counter, and the seven
            // waves that would only wait for the chain lanes at the first barrier put the request ~2 k cycles further ahead of
            // its use behind the matrix block - where the ISA showed a drained counter (vmcnt(0)) in every step.
            stash_load(a.H1, (long)t * ngroups + g, L, h1);
            // ... and with it the record of step t - 1 (used at the end of this step): requested behind the dz2 phase it was two more
            // loads in flight at the drain in the matrix block
            if (t > 0 && live) {
                const f32x4* rp = reinterpret_cast<const f32x4*>(a.SA + ((long)(t - 1) * R + tr) * SAW);
                const f32x4 r0 = rp[0], r1 = rp[1];
#pragma unroll
                for (int i = 0; i < 4; ++i) { rec_pre[i] = r0[i]; rec_pre[4 + i] = r1[i]; }
            }
            if (own) {
                float ga[2] = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 8; ++i) lam[i] = 0.f;
                float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, act[2] = {0.f, 0.f};
                if (live) {
#pragma unroll
                    for (int i = 0; i < OBS; ++i) o[i] = rec_cur[i];
#pragma unroll
                    for (int k = 0; k < ACT; ++k) act[k] = rec_cur[OBS + k];
                    if (t < a.n) {
                        float on[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int i = 0; i < OBS; ++i) on[i] = sCarry[tid * 16 + 8 + i];
                        float lam_next[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) lam_next[i] = sCarry[tid * 16 + i];
                        ENV::vjp(o, act, on, lam_next, sRho[t], lam, ga);
                    }
                    if ((selmask >> t) & 1u)             // (two of the 26 steps: the slice search stays a plain loop)
                    for (int ks = 0; ks < a.n_sel; ++ks)
                        if (a.sel[ks] == t) {
                            const float* gx = a.GXQ + ((long)ks * R + tr) * QIN;
#pragma unroll
                            for (int i = 0; i < OBS; ++i) lam[i] += gx[i] * a.obs_scale[i];
                            if constexpr (WIDE) {
                                if (t > 0) {
#pragma unroll
                                    for (int k = 0; k < MAXF; ++k)
                                        if (k < nf) lam[ENV::FUT_SRC] += gx[OBS + k] * a.obs_scale[OBS + k];
                                }
                            }
#pragma unroll
                            for (int k = 0; k < ACT; ++k) ga[k] += gx[OD + k];
                        }
                }
#pragma unroll
                for (int k = 0; k < ACT; ++k) {
                    float d = ga[k];
                    if (a.out_tanh) {
                        const float th = act[k] * a.inv_out_scale;
                        d *= a.out_scale * (1.f - th * th);
                    }
                    sD3[d3_index(tid, k)] = d;
                    if (live && a.DZ3 && (a.stash_all || t == 0))
                        a.DZ3[((long)(a.stash_all ? t : 0) * R + tr) * ACT + k] = d;
                    if constexpr (THIN) sB3[tid * 2 + k] += d;                       // db3 (rows beyond the batch carry d = 0)
                }
                if constexpr (THIN) {       // this step's network input of the row, as the forward sweep published it (obs * scale)
#pragma unroll
                    for (int i = 0; i < 8; ++i) sXin[xpar * GROUP * 8 + tid * 8 + i] = i < OBS ? o[i] * a.obs_scale[i] : 0.f;
                }
            }
            float dz1[2][4], dz2[2][4];
            lds_barrier();
            MPG_STAMP_AT(0);
            backward_dz2<NIN, ACT>(sD3, sA, L, r, h2_cur, dz2);
            if constexpr (THIN) {           // db2 += dz2, dW3 += h2 dz3^T: everything is at hand here (sD3 is stable until the step's last barrier)
                f32x4* th = reinterpret_cast<f32x4*>(sThin) + tid;
                f32x4 q0 = th[0];                                                    // gb1[0..1] gb2[0..1]
                float gw3[2][ACT];
                f32x4 q1 = th[NTHREAD];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int k = 0; k < ACT; ++k) gw3[tt][k] = q1[tt * ACT + k];
#pragma unroll
                for (int k = 0; k < ACT; ++k) {
                    const f32x4 d3 = *reinterpret_cast<const f32x4*>(sD3 + d3_index(4 * L.rg, k));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        gw3[0][k] = fmaf(h2_cur[0][j], d3[j], gw3[0][k]);
                        gw3[1][k] = fmaf(h2_cur[1][j], d3[j], gw3[1][k]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { q0[2] += dz2[0][j]; q0[3] += dz2[1][j]; }
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int k = 0; k < ACT; ++k) q1[tt * ACT + k] = gw3[tt][k];
                th[0] = q0;
                th[NTHREAD] = q1;
            }
            // all global loads of the step are issued HERE, behind the dz2 phase: h1 is consumed after the MFMA block,
            // the record and h2 stash of step t-1 in the next iteration (software pipeline)
            if (t > 0) {
                stash_load(a.H2, (long)(t - 1) * ngroups + g, L, h2_pre);
            }
            if (t > 0)
                backward_rest<NIN, ACT, true>(sD3, sA, sA1, sPartX, L, w2t, r, h1, dz1);
            else
                backward_rest<NIN, ACT, false>(sD3, sA, sA1, sPartX, L, w2t, r, h1, dz1);
            if (a.DZ2 && (a.stash_all || t == 0)) {
                const long sg = (long)(a.stash_all ? t : 0) * ngroups + g;
                if (!THIN) stash_store(a.DZ1, sg, L, dz1);          // (THIN: dz1 is consumed below, no stash)
                stash_store(a.DZ2, sg, L, dz2);
            }
            if (own) {
                if (t > 0) {
                    float dxr[XSW];
                    dx_reduce_row<XSW, WIDE ? XSW : OBS>(sPartX, tid, dxr);
#pragma unroll
                    for (int i = 0; i < OBS; ++i) lam[i] += dxr[i] * a.obs_scale[i];
                    if constexpr (WIDE) {       // t > 0 here: the observation of this step came out of the model
#pragma unroll
                        for (int k = 0; k < MAXF; ++k)
                            if (k < nf) lam[ENV::FUT_SRC] += dxr[OBS + k] * a.obs_scale[OBS + k];
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    sCarry[tid * 16 + i] = lam[i];
                    sCarry[tid * 16 + 8 + i] = rec_cur[i];
                    rec_cur[i] = rec_pre[i];
                }
            }
            if constexpr (THIN) {           // db1 += dz1, dW1 += x^T dz1 (behind the step's last barrier: the other seven waves
                                            // do this while wave 0's trajectory lanes run the serial chain)
                f32x4* th = reinterpret_cast<f32x4*>(sThin) + tid;
                f32x4 q0 = th[0];
#pragma unroll
                for (int j = 0; j < 4; ++j) { q0[0] += dz1[0][j]; q0[1] += dz1[1][j]; }
                th[0] = q0;
                float gw1[2][OBS];
                constexpr int BASE = 4 + 2 * ACT;                    // float index of gW1[0][0] in the lane's slots
#pragma unroll
                for (int e = 0; e < 2 * OBS; ++e) gw1[e / OBS][e % OBS] = sThin[((BASE + e) / 4 * NTHREAD + tid) * 4 + (BASE + e) % 4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float* xr = sXin + xpar * GROUP * 8 + L.row(j) * 8;
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(xr), x1 = *reinterpret_cast<const f32x4*>(xr + 4);
#pragma unroll
                    for (int i = 0; i < OBS; ++i) {
                        const float xv = i < 4 ? x0[i & 3] : x1[i & 3];
                        gw1[0][i] = fmaf(xv, dz1[0][j], gw1[0][i]);
                        gw1[1][i] = fmaf(xv, dz1[1][j], gw1[1][i]);
                    }
                }
#pragma unroll
                for (int e = 0; e < 2 * OBS; ++e) sThin[((BASE + e) / 4 * NTHREAD + tid) * 4 + (BASE + e) % 4] = gw1[e / OBS][e % OBS];
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int j = 0; j < 4; ++j) h2_cur[tt][j] = h2_pre[tt][j];
            xpar ^= 1;
            MPG_STAMP_AT(7);
            // next iteration: sD3 is rewritten by wave 0 only after it has passed backward_group's final barrier,
            // and read by the others only after the __syncthreads above -> no extra barrier needed.
        }
#ifdef MPG_STAMP
        if ((tid & 63) == 0 && a.dbg)
            for (int k = 0; k < 8; ++k) a.dbg[((long)blockIdx.x * NWAVE + (tid >> 6)) * 8 + k] = (float)g_st_acc[tid >> 6][k];
#endif
    }
    if constexpr (THIN) {
        // this workgroup's partial: the four row quads (lanes c, c + 16, c + 32, c + 48 of a wave) of every column summed in a fixed
        // order by the quad-0 lane, written in the network's flat layout without W2
        __syncthreads();
        if (L.rg == 0) {
            const int out_dim = 2 * ACT;
            float* dst = a.thin_part + (size_t)blockIdx.x * thin_floats(OBS, out_dim);
            float* dW1 = dst, *db1 = dW1 + OBS * H, *db2 = db1 + H, *dW3 = db2 + H, *db3 = dW3 + H * out_dim;
            auto acc = [&](int slot) {
                float s4 = 0.f;
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) s4 += sThin[((slot / 4) * NTHREAD + tid + 16 * rq) * 4 + slot % 4];
                return s4;
            };
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int col = L.col(tt);
                db1[col] = acc(tt);
                db2[col] = acc(2 + tt);
#pragma unroll
                for (int k = 0; k < ACT; ++k) dW3[col * out_dim + k] = acc(4 + tt * ACT + k);
#pragma unroll
                for (int k = ACT; k < 2 * ACT; ++k) dW3[col * out_dim + k] = 0.f;           // the unused log-std half (SURVEY B-5)
#pragma unroll
                for (int i = 0; i < OBS; ++i) dW1[i * H + col] = acc(4 + 2 * ACT + tt * OBS + i);
            }
            if (tid < 2 * ACT) {
                float s3 = 0.f;
                if (tid < ACT)
                    for (int row = 0; row < GROUP; ++row) s3 += sB3[row * 2 + tid];
                db3[tid] = s3;
            }
        }
    }
}

}  // namespace

// This file is compiled twice: as itself (path tracking: the bench's kernel, with the flags that suit it - mpg_amd/build.py) and,
// through rollout_bwd_pendulum.hip (MPG_BWD_PENDULUM_PART), for the pendulum instantiations, which measure 10 us slower under
// those flags and keep the previous ones.
#ifdef MPG_BWD_PENDULUM_PART
void launch_rollout_bwd_pendulum(const RollBwdArgs& ba, long ngroups, hipStream_t s) {
    if (ba.thin_part) hipLaunchKernelGGL((k_rollout_bwd<Pendulum, true, false, true>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, ba);
    else if (ba.pack) hipLaunchKernelGGL((k_rollout_bwd<Pendulum, true>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, ba);
    else hipLaunchKernelGGL((k_rollout_bwd<Pendulum, false>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, ba);
}
#else
void launch_rollout_bwd_pendulum(const RollBwdArgs& ba, long ngroups, hipStream_t s);

int launch_rollout_bwd(const RollBwdArgs& ba_in, int env_kind, long ngroups, int n, hipStream_t s, mpg_prof_t* prof) {
    RollBwdArgs ba = ba_in;
    ba.dbg = nullptr;
#ifdef MPG_STAMP
    static float* s_dbg_b = nullptr;
    static int s_calls_b = 0;
    if (!s_dbg_b) (void)hipMalloc(&s_dbg_b, 256 * 8 * 8 * sizeof(float));
    ba.dbg = s_dbg_b;
#endif
    mpg_prof_begin(prof, 1, s);
    if (env_kind == MPG_ENV_PATH_TRACKING && ba.obs_dim > PathTracking::OBS)
        { if (ba.pack) hipLaunchKernelGGL((k_rollout_bwd<PathTracking, true, true>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, ba); else hipLaunchKernelGGL((k_rollout_bwd<PathTracking, false, true>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, ba); }
    else if (env_kind == MPG_ENV_PATH_TRACKING && ba.thin_part)
        hipLaunchKernelGGL((k_rollout_bwd<PathTracking, true, false, true>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, ba);
    else if (env_kind == MPG_ENV_PATH_TRACKING)
        { if (ba.pack) hipLaunchKernelGGL((k_rollout_bwd<PathTracking, true>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, ba); else hipLaunchKernelGGL((k_rollout_bwd<PathTracking, false>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, ba); }
    else
        launch_rollout_bwd_pendulum(ba, ngroups, s);
    mpg_prof_end(prof, 1, s);
    MPG_CHECK_LAUNCH("k_rollout_bwd");
#ifdef MPG_STAMP
    if (++s_calls_b % 50 == 0) {
        static float h[256 * 8 * 8];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, s_dbg_b, sizeof(h), hipMemcpyDeviceToHost);
        const int nwg = grid_for(ngroups);
        for (int w = 0; w < 8; ++w) {
            double acc[8] = {0};
            for (int b = 0; b < nwg; ++b) for (int k = 0; k < 8; ++k) acc[k] += h[(b * 8 + w) * 8 + k];
            fprintf(stderr, "[stamp bwd] wave %d cycles/step:", w);
            double tot = 0;
            for (int k = 0; k < 8; ++k) { fprintf(stderr, " p%d=%.0f", k, acc[k] / nwg / (n + 1)); tot += acc[k] / nwg / (n + 1); }
            fprintf(stderr, " total=%.0f\n", tot);
        }
    }
#endif
    (void)n;
    return MPG_OK;
}
#endif   // MPG_BWD_PENDULUM_PART

}  // namespace rollout
