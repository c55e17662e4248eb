// Reverse sweep of the fused n-step model rollout (see rollout_kernels.hip for the overview and the references).
#include "rollout_common.h"

namespace rollout {
namespace {

template <class ENV, bool PK>
__global__ void __launch_bounds__(NTHREAD, 2) k_rollout_bwd(const RollBwdArgs a) {
    constexpr int OBS = ENV::OBS, ACT = ENV::ACT, QIN = OBS + ACT;
    __shared__ __attribute__((aligned(16))) float smem[2 * A_IMG + GROUP * MAXOUT + NWAVE * GROUP * XS];
    float* sA = smem;
    float* sA1 = sA + A_IMG;
    float* sD3 = sA1 + A_IMG;
    float* sPartX = sD3 + GROUP * MAXOUT;
    // Everything the 16 trajectory lanes read along the sweep is staged in LDS up front - the (obs | action) records of all
    // n + 1 steps, the critic-input gradients of the selected slices, dL/d(reward) per step: no global or scalar-memory
    // latency on the serial chain (a kernel-argument array indexed by the loop counter is a scalar load + wait per use), and
    // no registers held for the next step's record by 496 lanes that never use them.
    __shared__ __attribute__((aligned(16))) float sRec[(MAXN + 1) * GROUP * SAW];
    __shared__ __attribute__((aligned(16))) float sGX[MAXSEL * GROUP * SAW];
    __shared__ float sRho[MAXN];
    // The activation stashes reach the lanes through LDS-DMA (global_load_lds: no destination registers), requested a whole
    // step ahead: [2 buffers by step parity][wave][h1 tile 0, h1 tile 1, h2 tile 0, h2 tile 1][64 lanes x 16 B].  At n = 25 the
    // stash (218 MB) plus what the sweeps touch between writing and re-reading a line exceeds the 256 MiB Infinity Cache for about
    // half of the steps; their reads come from HBM, and a request made one matrix block (~0.7 us) before its use stalled every
    // such step (tools/roll_scale.py: the cost per step grows beyond n = 15 at B = 4096 and not at B = 2048).
    __shared__ __attribute__((aligned(16))) float sDma[2 * NWAVE * 4 * 256];
    static_assert(OBS + ACT <= SAW, "records hold obs | action in 8 floats");
    const Lane L;
    const int tid = threadIdx.x;
    const Net net = make_net(a.policy, OBS, 2 * ACT);
    float w2t[128];
    SmallRegs<OBS, ACT> r;
    if constexpr (PK) load_w2_packed(a.pack, L, w2t); else load_w2_bwd(net.W2, L, w2t);
    load_small<OBS, ACT>(net, L, r);
    const long R = (long)a.rows * a.M;
    const long ngroups = (R + GROUP - 1) / GROUP;
    {   // constant indices only: indexing a kernel-argument array by the thread id would move it to scratch memory
        float rv = 0.f;
#pragma unroll
        for (int i = 0; i < MAXN; ++i) rv = tid == i ? a.rho[i] : rv;
        if (tid < MAXN) sRho[tid] = rv;
    }
#ifdef MPG_STAMP
    if ((tid & 63) == 0) {
        for (int k = 0; k < 10; ++k) g_st_acc[tid >> 6][k] = 0;
        g_st_prev[tid >> 6] = __builtin_amdgcn_s_memtime();
    }
#endif
    for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const long tr = g * GROUP + tid;
        const bool own = tid < GROUP, live = own && tr < R;
        __syncthreads();                               // the previous group's reads of the staged arrays are done
        // request the stashes of step t into the buffer of its parity (this wave's 4 KB of it; the hardware adds lane * 16)
        auto request = [&](int t) {
            float* dst = sDma + (((t & 1) * NWAVE + L.wave) * 4) * 256;
            const long fo = (((long)t * ngroups + g) * 16 + 2 * L.wave) * 64 + L.lane;
            const f32x4* p1 = reinterpret_cast<const f32x4*>(a.H1) + fo;
            const f32x4* p2 = reinterpret_cast<const f32x4*>(a.H2) + fo;
            typedef __attribute__((address_space(3))) void* lds_t;
            __builtin_amdgcn_global_load_lds(p1, (lds_t)(dst), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(p1 + 64, (lds_t)(dst + 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(p2, (lds_t)(dst + 512), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(p2 + 64, (lds_t)(dst + 768), 16, 0, 0);
        };
        // the lane's G16 fragment pair of stash `which` (0: h1, 1: h2) of step t from the landed buffer
        auto landed = [&](int t, int which, float (&v)[2][4]) {
            const float* src = sDma + (((t & 1) * NWAVE + L.wave) * 4 + 2 * which) * 256 + 4 * L.lane;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(src + t2 * 256);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[t2][j] = q[j];
            }
        };
        request(a.n);                                  // before the staged arrays are waited for
        for (int idx = tid; idx < (a.n + 1) * GROUP * (SAW / 4); idx += NTHREAD) {       // one float4 per thread and pass
            const int t = idx / (GROUP * (SAW / 4)), rem = idx % (GROUP * (SAW / 4)), row = rem / (SAW / 4), q = rem % (SAW / 4);
            const long trj = g * GROUP + row;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (trj < R) v = reinterpret_cast<const f32x4*>(a.SA + ((long)t * R + trj) * SAW)[q];
            reinterpret_cast<f32x4*>(sRec)[idx] = v;
        }
        for (int idx = tid; idx < a.n_sel * GROUP * SAW; idx += NTHREAD) {
            const int ks = idx / (GROUP * SAW), row = (idx / SAW) % GROUP, i = idx % SAW;
            const long trj = g * GROUP + row;
            sGX[idx] = (trj < R && i < QIN) ? a.GXQ[((long)ks * R + trj) * QIN + i] : 0.f;
        }
        __syncthreads();
        float lam_next[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // trajectory lanes: dL/d(obs_{t+1})
        for (int t = a.n; t >= 0; --t) {
            if (own) {
                float lam[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ga[2] = {0.f, 0.f};
                float o[8], on[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, act[2] = {0.f, 0.f};
                {
                    const f32x4* rp = reinterpret_cast<const f32x4*>(sRec + ((t * GROUP) + tid) * SAW);
                    const f32x4 r0 = rp[0], r1 = rp[1];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { o[i] = r0[i]; o[4 + i] = r1[i]; }
                }
#pragma unroll
                for (int k = 0; k < ACT; ++k) act[k] = o[OBS + k];
#pragma unroll
                for (int i = OBS; i < 8; ++i) o[i] = 0.f;
                if (t < a.n) {
                    const f32x4* rp = reinterpret_cast<const f32x4*>(sRec + (((t + 1) * GROUP) + tid) * SAW);
                    const f32x4 r0 = rp[0], r1 = rp[1];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { on[i] = r0[i]; on[4 + i] = r1[i]; }
#pragma unroll
                    for (int i = OBS; i < 8; ++i) on[i] = 0.f;
                    ENV::vjp(o, act, on, lam_next, sRho[t], lam, ga);
                }
                // constant slice indices only (see the forward sweep)
#pragma unroll
                for (int ks = 0; ks < MAXSEL; ++ks)
                    if (ks < a.n_sel && a.sel[ks] == t) {
                        const float* gx = sGX + (ks * GROUP + tid) * SAW;
#pragma unroll
                        for (int i = 0; i < OBS; ++i) lam[i] += gx[i] * a.obs_scale[i];
#pragma unroll
                        for (int k = 0; k < ACT; ++k) ga[k] += gx[OBS + k];
                    }
#pragma unroll
                for (int k = 0; k < ACT; ++k) {
                    float d = live ? ga[k] : 0.f;
                    if (a.out_tanh) {
                        const float th = act[k] / a.out_scale;
                        d *= a.out_scale * (1.f - th * th);
                    }
                    sD3[d3_index(tid, k)] = d;
                    if (live && a.DZ3 && (a.stash_all || t == 0))
                        a.DZ3[((long)(a.stash_all ? t : 0) * R + tr) * ACT + k] = d;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) lam_next[i] = live ? lam[i] : 0.f;      // completed below by the policy's input gradient
            }
            float h1[2][4], h2[2][4], dz1[2][4], dz2[2][4];
            lds_barrier();
            MPG_STAMP_AT(0);
            // this step's stashes were requested a step ago: wait for them (vector-memory operations retire in order; nothing
            // younger is outstanding), then request the next step's into the other buffer
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (t > 0) request(t - 1);
            landed(t, 1, h2);
            backward_dz2<OBS, ACT>(sD3, sA, L, r, h2, dz2);
            landed(t, 0, h1);
            if (t > 0)
                backward_rest<OBS, ACT, true>(sD3, sA, sA1, sPartX, L, w2t, r, h1, dz1);
            else
                backward_rest<OBS, ACT, false>(sD3, sA, sA1, sPartX, L, w2t, r, h1, dz1);
            if (a.DZ1 && (a.stash_all || t == 0)) {
                const long sg = (long)(a.stash_all ? t : 0) * ngroups + g;
                stash_store(a.DZ1, sg, L, dz1);
                stash_store(a.DZ2, sg, L, dz2);
            }
            if (own && live && t > 0) {
                float dxr[XS];
                dx_reduce_row(sPartX, tid, dxr);
#pragma unroll
                for (int i = 0; i < OBS; ++i) lam_next[i] += dxr[i] * a.obs_scale[i];
            }
            MPG_STAMP_AT(7);
            // next iteration: sD3 is rewritten by wave 0 only after it has passed backward_rest's final barrier, and read by the
            // others only after the barrier above -> no extra barrier needed.
        }
#ifdef MPG_STAMP
        if ((tid & 63) == 0 && a.dbg)
            for (int k = 0; k < 8; ++k) a.dbg[((long)blockIdx.x * NWAVE + (tid >> 6)) * 8 + k] = (float)g_st_acc[tid >> 6][k];
#endif
    }
}

}  // namespace

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
    if (env_kind == MPG_ENV_PATH_TRACKING)
        { if (ba.pack) hipLaunchKernelGGL((k_rollout_bwd<PathTracking, true>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, ba); else hipLaunchKernelGGL((k_rollout_bwd<PathTracking, false>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, ba); }
    else
        { if (ba.pack) hipLaunchKernelGGL((k_rollout_bwd<Pendulum, true>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, ba); else hipLaunchKernelGGL((k_rollout_bwd<Pendulum, false>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, ba); }
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

}  // namespace rollout
