// Fused critic-side kernels: the clipped double-Q target, the critics' forward + error + backward, and the critic's
// value + input gradient at the rollout slices - each ONE launch in which a workgroup takes one 16-row group through
// several networks / directions, re-loading its register-stationary weights between phases.  At B = 4096 (one group
// per CU) every separate launch costs a prologue + ~3.5 us boundary for ~3.4 us of MFMA work; fusing removes 14 of the
// 22 launches of a gradient step.  Results are bit-identical to the unfused launchers (same device functions).
#include "mlp_wgrad.h"

namespace mlp {

namespace {

// PK (template parameter of every kernel here): the caller handed packed images for ALL its networks - the kernel then
// contains no strided loader at all (the uncached form costs registers the hot form cannot spare)
template <bool PK>
__device__ __forceinline__ void load_w2(const float* pack, const float* W2, bool bwd, const Lane& L, float (&w)[128]) {
    if constexpr (PK) load_w2_packed(pack, L, w);
    else if (bwd) load_w2_bwd(W2, L, w);
    else load_w2_fwd(W2, L, w);
}

// order of a phase's loads: the small per-lane pieces first, so that layer 1 and its barrier run while the 256 KB
// register image is still streaming in (the vector-memory counter retires in order)
#define MPG_UNPAREN(...) __VA_ARGS__
#define MPG_LOAD2(SMALL, IMAGE) do { MPG_UNPAREN SMALL; MPG_UNPAREN IMAGE; } while (0)


constexpr int SMEM_FLOATS = 2 * A_IMG + GROUP * XS + NWAVE * GROUP * MAXOUT + GROUP * MAXOUT + NWAVE * GROUP * XS + 4 * GROUP;
struct Smem {
    float *sA, *sA1, *sX, *sPart, *sD3, *sPartX, *sQ;
    __device__ explicit Smem(float* base) {
        sA = base;
        sA1 = sA + A_IMG;
        sX = sA1 + A_IMG;
        sPart = sX + GROUP * XS;
        sD3 = sPart + NWAVE * GROUP * MAXOUT;
        sPartX = sD3 + GROUP * MAXOUT;
        sQ = sPartX + NWAVE * GROUP * XS;
    }
};

// ---------------------------------------------------------------------------------------------------------------
struct TargetArgs {
    const float *pol, *q1, *q2;                 // target networks (q2 nullable)
    const float *pk_pol, *pk_q1, *pk_q2;        // packed forward images (nullable)
    int rows;
    const float *obs2, *rew, *smooth_eps;
    float scale[8];
    int out_tanh;
    float out_scale, sigma, clipc, rshift, rscale, gamma;
    float* y;
    // optional fused minibatch draw (ReplayBuffer.sample, buffer.py:70-78; same Philox stream as k_sample_gather): the
    // kernel's 16 rows are drawn and gathered by its own first 16 lanes and written out for the later kernels
    int draw, n_storage;      // draw: 0 none, 1 gather every row, 2 rows outside the fresh window are already in o_*
    int d_capacity, d_fresh_start, d_fresh_count;
    uint32_t dk0, dk1, dc1, dc2;
    const float *r_obs, *r_act, *r_rew, *r_obs2;
    const uint8_t* r_done;
    int* o_idx;
    float *o_obs, *o_act, *o_rew, *o_obs2, *o_done;
    int* status;               // nullable: MPG_STATUS_* word of the caller
    float* qpart;              // nullable: SPLIT launch (gridDim.y == 2) - workgroup (p, h) evaluates the target policy and target
                               // critic h only and leaves gamma-free Q values in qpart[h][rows]; the consumer takes the minimum
    unsigned long long* dbg;   // MPG_TIMELINE builds only
};

// G2 row groups per workgroup: every network's 256 KB register image is fetched once per workgroup and applied to G2 x 16
// rows.  With one group per workgroup the kernel is bound by the L2 -> CU traffic of the images (32 CUs of an XCD pull the
// same 256 KB at the same time: 64 B/clk each = the XCD's whole L2 bandwidth); two groups halve that traffic.
template <int OBS, int ACT, bool PK, int G2>
__global__ void __launch_bounds__(NTHREAD, 2) k_target_fused(const TargetArgs a) {
    constexpr int QIN = OBS + ACT;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    __shared__ __attribute__((aligned(16))) float sXg[G2][GROUP * XS];
    __shared__ float sQg[G2][2 * GROUP];
    const Smem m(smem);
    const Lane L;
    const int tid = threadIdx.x;
    const long g0 = (long)blockIdx.x * G2;
    // split launch: two workgroups per pair of row groups, one per target critic (both evaluate the target policy: a third of
    // the passes twice instead of half of the chip idle); only half 0 writes the drawn minibatch
    const int half = a.qpart ? (int)blockIdx.y : -1;
    __shared__ float sRew[G2 * GROUP];
    MPG_TL_DECL
    MPG_TL(0);
    // Minibatch draw: the trajectory lanes issue their (random-access) ring reads FIRST, then every wave requests the
    // policy's weights, and only then are the gathered values consumed - the vector-memory counter retires in order, so
    // this is the order in which the two latencies overlap instead of adding up.
    float o1[OBS], o2[OBS], ac[ACT], rw = 0.f;
    uint8_t dn = 0;
    long sr = 0;
    bool saw_nan = false;
#pragma unroll
    for (int i = 0; i < OBS; ++i) o1[i] = o2[i] = 0.f;
#pragma unroll
    for (int k = 0; k < ACT; ++k) ac[k] = 0.f;
    const bool drawn = a.draw && tid < G2 * GROUP && g0 * GROUP + tid < a.rows;
    bool gather = drawn;
    if (drawn) {
        const long gr = g0 * GROUP + tid;
        const Philox4 p = philox4x32_10((uint32_t)(gr >> 2), a.dc1, a.dc2, 0x1d5u, a.dk0, a.dk1);
        sr = (long)(((uint64_t)philox_word(p, (int)(gr & 3)) * (uint64_t)a.n_storage) >> 32);
        if (a.draw == 2) {     // gathered ahead of time by the env launch unless the slot was being written then
            int off = (int)sr - a.d_fresh_start;
            if (off < 0) off += a.d_capacity;
            gather = off < a.d_fresh_count;
        }
        if (gather) {
#pragma unroll
            for (int i = 0; i < OBS; ++i) { o1[i] = a.r_obs[sr * OBS + i]; o2[i] = a.r_obs2[sr * OBS + i]; }
#pragma unroll
            for (int k = 0; k < ACT; ++k) ac[k] = a.r_act[sr * ACT + k];
            rw = a.r_rew[sr];
            dn = a.r_done[sr];
        } else {
#pragma unroll
            for (int i = 0; i < OBS; ++i) o2[i] = a.o_obs2[gr * OBS + i];
            rw = a.o_rew[gr];
        }
    }
    float w2[128], h1[2][4], h2[2][4];
    float zmax = 0.f;                          // largest first-layer activation seen by this lane (the engine's envelope, mlp_core.h)
    const Net pnet = make_net(a.pol, OBS, 2 * ACT);
    // output biases and smoothing noise of the output threads, requested up front (at their point of use each is a memory
    // round trip between two passes)
    float pb3 = 0.f, qb3[2] = {0.f, 0.f}, sm_eps = 0.f;
    if (tid < G2 * GROUP * ACT) {
        pb3 = pnet.b3[tid % ACT];
        const long gr0 = (g0 + tid / (GROUP * ACT)) * GROUP + (tid / ACT) % GROUP;
        if (a.smooth_eps && gr0 < a.rows) sm_eps = a.smooth_eps[gr0 * ACT + tid % ACT];
    }
    if (tid < G2 * GROUP) {
        qb3[0] = make_net(a.q1, OBS + ACT, 1).b3[0];
        if (a.q2) qb3[1] = make_net(a.q2, OBS + ACT, 1).b3[0];
    }
    SmallRegs<OBS, ACT> pr;
    MPG_TL(1);
    MPG_LOAD2((load_small<OBS, ACT>(pnet, L, pr)), (load_w2<PK>(a.pk_pol, pnet.W2, false, L, w2)));
    MPG_TL(2);
    if (a.draw) {
        if (tid < G2 * GROUP) {
            if (gather && half <= 0) {
                const long gr = g0 * GROUP + tid;
                if (a.o_idx) a.o_idx[gr] = (int)sr;
#pragma unroll
                for (int i = 0; i < OBS; ++i) { a.o_obs[gr * OBS + i] = o1[i]; a.o_obs2[gr * OBS + i] = o2[i]; }
#pragma unroll
                for (int k = 0; k < ACT; ++k) a.o_act[gr * ACT + k] = ac[k];
                a.o_rew[gr] = rw;
                if (a.o_done) a.o_done[gr] = (float)dn;
            }
            sRew[tid] = rw;
            saw_nan |= rw != rw;
#pragma unroll
            for (int i = 0; i < XS; ++i) sXg[tid / GROUP][(tid % GROUP) * XS + i] = i < OBS ? o2[i < OBS ? i : 0] * a.scale[i] : 0.f;
#pragma unroll
            for (int i = 0; i < OBS; ++i) saw_nan |= o2[i] != o2[i];
        }
    } else if (tid < G2 * GROUP * XS) {
        const int g2 = tid / (GROUP * XS), e = tid % (GROUP * XS), row = e / XS, i = e % XS;
        const long gr = (g0 + g2) * GROUP + row;
        const float xin = (gr < a.rows && i < OBS) ? a.obs2[gr * OBS + i] * a.scale[i] : 0.f;
        sXg[g2][e] = xin;
        saw_nan |= xin != xin;
        if (i == 0) {
            const float rin = gr < a.rows ? a.rew[gr] : 0.f;
            sRew[g2 * GROUP + row] = rin;
            saw_nan |= rin != rin;
        }
    }
    MPG_TL(3);
    lds_barrier();
    MPG_TL(4);
    float h1b[2][4], h2b[2][4];
    // G2 == 2: both row groups go through every network as a pair (forward_group2: one pair of barriers for two groups)
    if constexpr (G2 == 2) {   // a' = pi_target(s~')  (+ clip(sigma*eps, +-c), td3.py:74-76)
        forward_group2<OBS, ACT>(sXg[0], sXg[1], m.sA, m.sA1, m.sPart, m.sPartX, L, w2, pr, h1, h2, h1b, h2b, &zmax);
    } else {
        forward_group<OBS, ACT>(sXg[0], m.sA, m.sPart, L, w2, pr, h1, h2, nullptr, 0, nullptr, &zmax);
    }
    if (tid < G2 * GROUP * ACT) {
        const int g2 = tid / (GROUP * ACT), row = (tid / ACT) % GROUP, k = tid % ACT;
        const long gr = (g0 + g2) * GROUP + row;
        const float z = out_preact(g2 == 0 ? m.sPart : m.sPartX, pb3, row, k);
        float act = a.out_tanh ? a.out_scale * tanhf(z) : z;
        if (a.smooth_eps && gr < a.rows) act += fminf(fmaxf(a.sigma * sm_eps, -a.clipc), a.clipc);
        sXg[g2][row * XS + OBS + k] = act;
    }
    MPG_TL(5);
    lds_barrier();
    MPG_TL(6);
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        const float* qp = qi == 0 ? a.q1 : a.q2;
        if (!qp) break;
        if (half >= 0 && qi != half) continue;          // (workgroup-uniform)
        const Net net = make_net(qp, QIN, 1);
        SmallRegs<QIN, 1> r;
        MPG_LOAD2((load_small<QIN, 1>(net, L, r)), (load_w2<PK>(qi == 0 ? a.pk_q1 : a.pk_q2, net.W2, false, L, w2)));
        MPG_TL(10 + 6 * qi);
        if constexpr (G2 == 2) {
            forward_group2<QIN, 1>(sXg[0], sXg[1], m.sA, m.sA1, m.sPart, m.sPartX, L, w2, r, h1, h2, h1b, h2b, &zmax);
        } else {
            forward_group<QIN, 1>(sXg[0], m.sA, m.sPart, L, w2, r, h1, h2, nullptr, 0, nullptr, &zmax);
        }
        if (tid < G2 * GROUP) {
            const int g2 = tid / GROUP, row = tid % GROUP;
            sQg[g2][qi * GROUP + row] = out_preact(g2 == 0 ? m.sPart : m.sPartX, qb3[qi], row, 0);
        }
        MPG_TL(11 + 6 * qi);
        lds_barrier();
        MPG_TL(12 + 6 * qi);
    }
    MPG_TL(22);
    if (tid < G2 * GROUP) {
        const int g2 = tid / GROUP, row = tid % GROUP;
        const long gr = (g0 + g2) * GROUP + row;
        if (gr < a.rows) {
            const float pz = row_poison(sXg[g2] + row * XS, OBS);               // NaN in s': the target is NaN (see row_poison)
            if (half >= 0) a.qpart[(long)half * a.rows + gr] = sQg[g2][half * GROUP + row] + pz;
            else {
                const float q = a.q2 ? fminf(sQg[g2][row], sQg[g2][GROUP + row]) : sQg[g2][row];
                a.y[gr] = (sRew[tid] + a.rshift) * a.rscale + a.gamma * q + pz;      // mpg_learner.py:132-133
            }
        }
    }
    report_activation_range(a.status, zmax);
    report_nan(a.status, saw_nan);
    MPG_TL(23);
    MPG_TL_DUMP(a.dbg);
}

// ---------------------------------------------------------------------------------------------------------------
struct QlossArgs {
    const float* q[2];
    const float *pkf[2], *pkb[2];
    int rows;
    XSpec x;
    const float* y;
    float inv_b;
    CriticStash st[2];
    float* loss_part;          // [n_q][ngroups]
    float* td;
    int* status;               // nullable: MPG_STATUS_* word of the caller
    // k_critic_fused only, nullable: the target launch ran in its split form - y = (rew + shift) * scale + gamma * min(qpart[0],
    // qpart[1]) is finished here (the same arithmetic as k_target_fused's) and written to y_out by the Q1 workgroups
    const float* qpart;
    const float* rew;
    float rshift, rscale, gamma;
    float* y_out;
};

template <int QIN, bool PK>
__global__ void __launch_bounds__(NTHREAD, 2) k_qloss_fused(const QlossArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    const Smem m(smem);
    const Lane L;
    const int tid = threadIdx.x;
    const long g = blockIdx.x;
    const int qi = blockIdx.y;
    const long ngroups = gridDim.x;
    const Net net = make_net(a.q[qi], QIN, 1);
    const CriticStash st = a.st[qi];
    load_x_group<QIN>(a.x, a.rows, g, m.sX);
    lds_barrier();
    float w2[128], h1[2][4], h2[2][4], dz1[2][4], dz2[2][4];
    float zmax = 0.f;
    SmallRegs<QIN, 1> r;
    MPG_LOAD2((load_small<QIN, 1>(net, L, r)), (load_w2<PK>(a.pkf[qi], net.W2, false, L, w2)));
    forward_group<QIN, 1>(m.sX, m.sA, m.sPart, L, w2, r, h1, h2, nullptr, 0, nullptr, &zmax);
    report_activation_range(a.status, zmax);
    stash_store(st.h1, g, L, h1);
    stash_store(st.h2, g, L, h2);
    if (tid < GROUP) {   // err = Q(s~,a) - y; dL/dq = err / B_global   (mpg_learner.py:331-336)
        const long gr = g * GROUP + tid;
        float e = 0.f;
        if (gr < a.rows) {
            e = out_preact(m.sPart, net.b3[0], tid, 0) - a.y[gr] + row_poison(m.sX + tid * XS, QIN);
            st.dz3[gr] = e * a.inv_b;
            if (a.td && qi == 0) a.td[gr] = e;
        }
        m.sD3[d3_index(tid, 0)] = e * a.inv_b;
        m.sQ[tid] = e * e;
        report_nan(a.status, e != e);
    }
    MPG_TL(4);
    load_w2<PK>(a.pkb[qi], net.W2, true, L, w2);     // same registers, backward image
    lds_barrier();
    MPG_TL(5);
    if (tid == 0) {
        float s2 = 0.f;
        for (int i = 0; i < GROUP; ++i) s2 += m.sQ[i];
        a.loss_part[qi * ngroups + g] = 0.5f * a.inv_b * s2;
    }
    backward_group<QIN, 1, false>(m.sD3, m.sA, m.sA1, m.sPartX, L, w2, r, h1, h2, dz1, dz2);
    stash_store(st.dz1, g, L, dz1);
    stash_store(st.dz2, g, L, dz2);
}

// ---------------------------------------------------------------------------------------------------------------
struct QsliceArgs {
    const float* q;
    const float *pkf, *pkb;
    int R, n_sel;              // rows per slice, slices; total rows n_sel * R, one group per workgroup
    const float *xq, *gk;
    float gpow[4], coef[4];
    float* ret_part;           // [ngroups_total][2]: sum and sum of squares of G + gpow*q over the group's rows
    float* gxq;
    int* status;               // nullable: MPG_STATUS_* word of the caller
};

template <int QIN, bool PK>
__global__ void __launch_bounds__(NTHREAD, 2) k_qslice_fused(const QsliceArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    const Smem m(smem);
    const Lane L;
    const int tid = threadIdx.x;
    const long g = blockIdx.x;
    const long total = (long)a.n_sel * a.R;
    const Net net = make_net(a.q, QIN, 1);
    if (tid < GROUP * XS) {
        const int row = tid / XS, i = tid % XS;
        const long gr = g * GROUP + row;
        const float xin = (gr < total && i < QIN) ? a.xq[gr * QIN + i] : 0.f;
        m.sX[tid] = xin;
        report_nan(a.status, xin != xin);
    }
    lds_barrier();
    float w2[128], h1[2][4], h2[2][4], dz1[2][4], dz2[2][4];
    float zmax = 0.f;
    SmallRegs<QIN, 1> r;
    MPG_LOAD2((load_small<QIN, 1>(net, L, r)), (load_w2<PK>(a.pkf, net.W2, false, L, w2)));
    forward_group<QIN, 1>(m.sX, m.sA, m.sPart, L, w2, r, h1, h2, nullptr, 0, nullptr, &zmax);
    report_activation_range(a.status, zmax);
    if (tid < GROUP) {
        const long gr = g * GROUP + tid;
        float ret = 0.f, d = 0.f;
        if (gr < total) {
            const int k = (int)(gr / a.R);                               // slice of this row (R % 16 == 0 is not required)
            ret = a.gk[gr] + a.gpow[k] * out_preact(m.sPart, net.b3[0], tid, 0);   // mpg_learner.py:266
            d = a.coef[k];
        }
        m.sD3[d3_index(tid, 0)] = d;
        m.sQ[tid] = ret;
        m.sQ[GROUP + tid] = (gr < total) ? (float)(gr / a.R) : -1.f;
    }
    load_w2<PK>(a.pkb, net.W2, true, L, w2);
    lds_barrier();
    if (tid == 0) {   // a group may straddle two slices only when R % 16 != 0; the launcher requires R % 16 == 0
        float s1 = 0.f, s2 = 0.f;
        for (int i = 0; i < GROUP; ++i) { s1 += m.sQ[i]; s2 += m.sQ[i] * m.sQ[i]; }
        a.ret_part[g * 2] = s1;
        a.ret_part[g * 2 + 1] = s2;
    }
    backward_group<QIN, 1, true>(m.sD3, m.sA, m.sA1, m.sPartX, L, w2, r, h1, h2, dz1, dz2);
    if (tid < GROUP * QIN) {
        const int row = tid / QIN, i = tid % QIN;
        const long gr = g * GROUP + row;
        if (gr < total) a.gxq[gr * QIN + i] = dx_reduce(m.sPartX, row, i);
    }
}

// Two slices per workgroup (the default {0, n} selection): workgroup gb handles row group gb of BOTH slices, so each of
// the two register images of W2 is loaded once per 32 rows instead of once per 16 (the prologue, not the MFMA work,
// dominates these kernels at one row group per CU).
template <int QIN, bool PK>
__global__ void __launch_bounds__(NTHREAD, 2) k_qslice_fused2(const QsliceArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    __shared__ __attribute__((aligned(16))) float sX2[2 * GROUP * XS];
    __shared__ __attribute__((aligned(16))) float sD32[2 * GROUP * MAXOUT];
    __shared__ float sQ2[2 * GROUP];
    const Smem m(smem);
    const Lane L;
    const int tid = threadIdx.x;
    const long gpers = a.R / GROUP;                    // groups per slice (R % 16 == 0)
    const long gb = blockIdx.x;
    const Net net = make_net(a.q, QIN, 1);
    if (tid < 2 * GROUP * XS) {
        const int sl = tid / (GROUP * XS), row = (tid / XS) % GROUP, i = tid % XS;
        const long gr = (sl * gpers + gb) * GROUP + row;
        const float xin = i < QIN ? a.xq[gr * QIN + i] : 0.f;
        sX2[tid] = xin;
        report_nan(a.status, xin != xin);
    }
    lds_barrier();
    float w2[128], h1[2][2][4], h2[2][2][4], dz1[2][4], dz2[2][4];
    float zmax = 0.f;
    SmallRegs<QIN, 1> r;
    MPG_LOAD2((load_small<QIN, 1>(net, L, r)), (load_w2<PK>(a.pkf, net.W2, false, L, w2)));
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        forward_group<QIN, 1>(sX2 + sl * GROUP * XS, m.sA, m.sPart, L, w2, r, h1[sl], h2[sl], nullptr, 0, nullptr, &zmax);
        if (tid < GROUP) {
            const long gr = (sl * gpers + gb) * GROUP + tid;
            sQ2[sl * GROUP + tid] = a.gk[gr] + a.gpow[sl] * out_preact(m.sPart, net.b3[0], tid, 0);   // mpg_learner.py:266
            sD32[sl * GROUP * MAXOUT + d3_index(tid, 0)] = a.coef[sl];
        }
    }
    report_activation_range(a.status, zmax);
    load_w2<PK>(a.pkb, net.W2, true, L, w2);
    lds_barrier();
    if (tid < 2) {
        float s1 = 0.f, s2 = 0.f;
        for (int i = 0; i < GROUP; ++i) { const float q = sQ2[tid * GROUP + i]; s1 += q; s2 += q * q; }
        a.ret_part[(tid * gpers + gb) * 2] = s1;
        a.ret_part[(tid * gpers + gb) * 2 + 1] = s2;
    }
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        backward_group<QIN, 1, true>(sD32 + sl * GROUP * MAXOUT, m.sA, m.sA1, m.sPartX, L, w2, r, h1[sl], h2[sl], dz1, dz2);
        if (tid < GROUP * QIN) {
            const int row = tid / QIN, i = tid % QIN;
            const long gr = (sl * gpers + gb) * GROUP + row;
            a.gxq[gr * QIN + i] = dx_reduce(m.sPartX, row, i);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Critic losses AND the critic at the two selected rollout slices in one launch (the default {0, n} selection):
// workgroup (g, 0) runs Q1 on row group g of the replay batch and of both slices - three forward passes behind ONE load of
// the forward register image, three reverse passes behind one load of the transposed image - workgroup (g, 1) runs Q2 on
// the batch group.  Same arithmetic as k_qloss_fused + k_qslice_fused2, two weight-image loads per CU and one launch less.
struct CriticArgs {
    QlossArgs ql;
    QsliceArgs qs;
    unsigned long long* dbg;   // MPG_TIMELINE builds only
};

template <int QIN, bool PK>
__global__ void __launch_bounds__(NTHREAD, 2) k_critic_fused(const CriticArgs ca) {
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    __shared__ __attribute__((aligned(16))) float sX2[2 * GROUP * XS];
    __shared__ __attribute__((aligned(16))) float sD32[2 * GROUP * MAXOUT];
    __shared__ float sQ2[2 * GROUP];
    const QlossArgs& a = ca.ql;
    const QsliceArgs& q = ca.qs;
    const Smem m(smem);
    const Lane L;
    const int tid = threadIdx.x;
    const long g = blockIdx.x;
    const int qi = blockIdx.y;
    const long ngroups = gridDim.x;                    // == q.R / GROUP == a.rows / GROUP
    const bool slices = qi == 0;
    MPG_TL(0);
    const Net net = make_net(a.q[qi], QIN, 1);
    const CriticStash st = a.st[qi];
    // The group's inputs are REQUESTED first, then the small pieces and the 256 KB image, and only then are the inputs
    // consumed (LDS + barrier): the vector-memory counter retires in order, so this is the order in which the input round
    // trip (2 us in the first workgroup of a CU, cold) overlaps the image request instead of preceding it.
    float xv = 0.f, xv2 = 0.f;
    if (tid < GROUP * XS) {
        const int row = tid / XS, i = tid % XS;
        const long gr = g * GROUP + row;
        if (gr < a.rows && i < QIN) xv = i < a.x.d0 ? a.x.x0[gr * a.x.ld0 + i] * a.x.scale[i] : a.x.x1[gr * a.x.ld1 + (i - a.x.d0)];
    }
    if (slices && tid < 2 * GROUP * XS) {
        const int sl = tid / (GROUP * XS), row = (tid / XS) % GROUP, i = tid % XS;
        const long gr = (sl * ngroups + g) * GROUP + row;
        xv2 = i < QIN ? q.xq[gr * QIN + i] : 0.f;
    }
    // ... and so is everything the 16 output lanes will need behind the forward passes (the target's inputs, the slices'
    // reward sums, the output bias): loaded where they are used they are a memory round trip in the middle of the kernel, and
    // the wait in front of them (vmcnt counts stores too) also drains the stash stores issued just before
    float t_y = 0.f, t_rew = 0.f, t_q1 = 0.f, t_q2 = 0.f, gk_in[2] = {0.f, 0.f};
    const float b3v = net.b3[0];
    if (tid < GROUP) {
        const long gr = g * GROUP + tid;
        if (gr < a.rows) {
            if (a.qpart) { t_rew = a.rew[gr]; t_q1 = a.qpart[gr]; t_q2 = a.qpart[(long)a.rows + gr]; }
            else t_y = a.y[gr];
        }
        if (slices) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) gk_in[sl] = q.gk[(sl * ngroups + g) * GROUP + tid];
        }
    }
    float w2[128], h1[3][2][4], h2[3][2][4], dz1[2][4], dz2[2][4];
    float zmax = 0.f;
    SmallRegs<QIN, 1> r;
    MPG_LOAD2((load_small<QIN, 1>(net, L, r)), (load_w2<PK>(a.pkf[qi], net.W2, false, L, w2)));
    if (tid < GROUP * XS) m.sX[tid] = xv;
    if (slices && tid < 2 * GROUP * XS) sX2[tid] = xv2;
    report_nan(a.status, xv2 != xv2);           // the critic inputs at the rollout slices (their NaN already sits in G_k and the adjoint)
    lds_barrier();
    MPG_TL(1);
    MPG_TL(2);
    // ---- forward: replay batch group ----
    forward_group<QIN, 1>(m.sX, m.sA, m.sPart, L, w2, r, h1[0], h2[0], nullptr, 0, nullptr, &zmax);
    MPG_TL(3);
    stash_store(st.h1, g, L, h1[0]);
    stash_store(st.h2, g, L, h2[0]);
    if (tid < GROUP) {   // err = Q(s~,a) - y; dL/dq = err / B_global   (mpg_learner.py:331-336)
        const long gr = g * GROUP + tid;
        float e = 0.f;
        if (gr < a.rows) {
            float yv = t_y;
            if (a.qpart) {
                yv = (t_rew + a.rshift) * a.rscale + a.gamma * fminf(t_q1, t_q2);   // mpg_learner.py:132-133
                if (qi == 0) a.y_out[gr] = yv;
            }
            e = out_preact(m.sPart, b3v, tid, 0) - yv + row_poison(m.sX + tid * XS, QIN);
            st.dz3[gr] = e * a.inv_b;
            if (a.td && qi == 0) a.td[gr] = e;
        }
        m.sD3[d3_index(tid, 0)] = e * a.inv_b;
        m.sQ[tid] = e * e;
        report_nan(a.status, e != e);           // a NaN in (s, a), in the target or in the reward of this row
    }
    // ---- forward: the two slices (Q1 workgroups only; wave-uniform branch) ----
    if (slices) {
        // the two slices as a pair behind one pair of barriers (forward_group2: the same arithmetic per group).  m.sPart of the
        // batch group was read by its 16 output lanes above, in front of the barriers inside
        forward_group2<QIN, 1>(sX2, sX2 + GROUP * XS, m.sA, m.sA1, m.sPart, m.sPartX, L, w2, r, h1[1], h2[1], h1[2], h2[2], &zmax);
        if (tid < GROUP) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                sQ2[sl * GROUP + tid] = gk_in[sl] + q.gpow[sl] * out_preact(sl == 0 ? m.sPart : m.sPartX, b3v, tid, 0);   // mpg_learner.py:266
                sD32[sl * GROUP * MAXOUT + d3_index(tid, 0)] = q.coef[sl];
            }
        }
    }
    MPG_TL(4);
    report_activation_range(a.status, zmax);
    load_w2<PK>(a.pkb[qi], net.W2, true, L, w2);     // same registers, backward image
    lds_barrier();
    MPG_TL(5);
    if (tid == 0) {
        float s2 = 0.f;
        for (int i = 0; i < GROUP; ++i) s2 += m.sQ[i];
        a.loss_part[qi * ngroups + g] = 0.5f * a.inv_b * s2;
    }
    if (slices && tid >= 64 && tid < 66) {
        const int sl = tid - 64;
        float s1 = 0.f, s2 = 0.f;
        for (int i = 0; i < GROUP; ++i) { const float v = sQ2[sl * GROUP + i]; s1 += v; s2 += v * v; }
        q.ret_part[(sl * ngroups + g) * 2] = s1;
        q.ret_part[(sl * ngroups + g) * 2 + 1] = s2;
    }
    // ---- reverse: replay batch group (weight-gradient stashes), then the slices (input gradients) ----
    backward_group<QIN, 1, false>(m.sD3, m.sA, m.sA1, m.sPartX, L, w2, r, h1[0], h2[0], dz1, dz2);
    stash_store(st.dz1, g, L, dz1);
    stash_store(st.dz2, g, L, dz2);
    MPG_TL(6);
    if (slices) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            backward_group<QIN, 1, true>(sD32 + sl * GROUP * MAXOUT, m.sA, m.sA1, m.sPartX, L, w2, r, h1[1 + sl], h2[1 + sl], dz1, dz2);
            if (tid < GROUP * QIN) {
                const int row = tid / QIN, i = tid % QIN;
                const long gr = (sl * ngroups + g) * GROUP + row;
                q.gxq[gr * QIN + i] = dx_reduce(m.sPartX, row, i);
            }
        }
    }
    MPG_TL(7);
#ifdef MPG_TIMELINE
    __syncthreads();
    if (ca.dbg && blockIdx.x == 7 && threadIdx.x < NWAVE * MPG_TL_MARKS)
        ca.dbg[blockIdx.y * NWAVE * MPG_TL_MARKS + threadIdx.x] = s_tl[threadIdx.x / MPG_TL_MARKS][threadIdx.x % MPG_TL_MARKS];
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// The same launch BALANCED over the chip (round 4).  k_critic_fused gives a CU one Q1 workgroup (three forward and three reverse
// passes behind two image loads) and then one Q2 workgroup (one + one pass behind two more image loads): four 256 KB image loads
// in sequence per CU, ~9k cycles each - more than half of the kernel.  Here every workgroup takes FOUR row groups of ONE kind -
// 3/4 of the workgroups the critic Q1 (replay-batch groups, then the groups of slice 0, then those of slice n), 1/4 the critic Q2
// (replay-batch groups) - so a CU runs one workgroup: eight passes behind TWO image loads.  The four forward passes run as two
// pairs (forward_group2), the reverse passes one by one.  Same device functions on the same operands as k_critic_fused:
// bit-identical stashes, targets, losses, returns and input gradients.  Needs rows / 16 divisible by 4.
template <int QIN, bool PK>
__global__ void __launch_bounds__(NTHREAD, 2) k_critic_fused4(const CriticArgs ca) {
    constexpr int G4 = 4;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    __shared__ __attribute__((aligned(16))) float sX4[G4 * GROUP * XS];
    __shared__ __attribute__((aligned(16))) float sD34[G4 * GROUP * MAXOUT];
    __shared__ float sQ4[G4 * GROUP];
    __shared__ __attribute__((aligned(16))) float sA2[A_IMG];                 // the second dz2 image of a reverse pair
    __shared__ __attribute__((aligned(16))) float sPartX2[NWAVE * GROUP * XS];
    // h1 / h2 of the FIRST forward pair wait here for their reverse pass (64 KB; holding all four groups in registers spills):
    // float4 index (slot * 512 + thread), slot = (group of the pair * 2 + layer) * 2 + tile - every lane re-reads what it wrote
    __shared__ __attribute__((aligned(16))) float sHold[8 * NTHREAD * 4];
    const QlossArgs& a = ca.ql;
    const QsliceArgs& q = ca.qs;
    const Smem m(smem);
    const Lane L;
    const int tid = threadIdx.x;
    const long ngroups = a.rows / GROUP;                       // row groups of the replay batch == of each slice
    const long nq1 = 3 * ngroups / G4;                         // workgroups of Q1: units [0, 3 ngroups) = replay | slice 0 | slice n
    const bool q2wg = (long)blockIdx.x >= nq1;
    const int qi = q2wg ? 1 : 0;
    const long u0 = (q2wg ? (long)blockIdx.x - nq1 : (long)blockIdx.x) * G4;
    const int kind = q2wg ? 0 : (int)(u0 / ngroups);           // 0: replay batch, 1: slice 0, 2: slice n (workgroup-uniform)
    const long g0 = u0 - (long)kind * ngroups;                 // the workgroup's groups: g0 .. g0 + 3 of that kind
    const bool slices = kind > 0;
    const int sl = kind > 0 ? kind - 1 : 0;
    const Net net = make_net(a.q[qi], QIN, 1);
    const CriticStash st = a.st[qi];
    // inputs of the four groups: one (row, column) per thread, requested ahead of the weight image (see k_critic_fused)
    float xv = 0.f;
    {
        const int u4 = tid / (GROUP * XS), e = tid % (GROUP * XS), row = e / XS, i = e % XS;
        const long gr = (g0 + u4) * GROUP + row;
        if (i < QIN) {
            if (slices) xv = q.xq[((long)sl * a.rows + gr) * QIN + i];
            else xv = i < a.x.d0 ? a.x.x0[gr * a.x.ld0 + i] * a.x.scale[i] : a.x.x1[gr * a.x.ld1 + (i - a.x.d0)];
        }
    }
    float t_y = 0.f, t_rew = 0.f, t_q1 = 0.f, t_q2 = 0.f, gk_in = 0.f;
    const float b3v = net.b3[0];
    if (tid < G4 * GROUP) {
        const long gr = (g0 + tid / GROUP) * GROUP + tid % GROUP;
        if (slices) gk_in = q.gk[(long)sl * a.rows + gr];
        else if (a.qpart) { t_rew = a.rew[gr]; t_q1 = a.qpart[gr]; t_q2 = a.qpart[(long)a.rows + gr]; }
        else t_y = a.y[gr];
    }
    float w2[128], h1[2][2][4], h2[2][2][4], dz1[2][4], dz2[2][4];      // h1 / h2: the pair in flight
    float zmax = 0.f;
    SmallRegs<QIN, 1> r;
    MPG_LOAD2((load_small<QIN, 1>(net, L, r)), (load_w2<PK>(a.pkf[qi], net.W2, false, L, w2)));
    sX4[tid] = xv;
    if (slices) report_nan(a.status, xv != xv);
    lds_barrier();
    // ---- forward: two pairs of groups behind one image ----
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        forward_group2<QIN, 1>(sX4 + (2 * p) * GROUP * XS, sX4 + (2 * p + 1) * GROUP * XS, m.sA, m.sA1, m.sPart, m.sPartX, L, w2, r,
                               h1[0], h2[0], h1[1], h2[1], &zmax);
        if (!slices) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                stash_store(st.h1, g0 + 2 * p + e, L, h1[e]);
                stash_store(st.h2, g0 + 2 * p + e, L, h2[e]);
            }
        }
        if (p == 0) {       // park the first pair's activations in LDS; the second pair's stay in registers for the reverse passes
            f32x4* hold = reinterpret_cast<f32x4*>(sHold) + tid;
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    hold[((e * 2 + 0) * 2 + t) * NTHREAD] = f32x4{h1[e][t][0], h1[e][t][1], h1[e][t][2], h1[e][t][3]};
                    hold[((e * 2 + 1) * 2 + t) * NTHREAD] = f32x4{h2[e][t][0], h2[e][t][1], h2[e][t][2], h2[e][t][3]};
                }
        }
        if (tid < G4 * GROUP && (tid >> 5) == p) {          // the 32 output lanes of this pair (wave 0: ordered against the next pair by its barrier)
            const int u4 = tid / GROUP, row = tid % GROUP;
            const long gr = (g0 + u4) * GROUP + row;
            const float qv = out_preact((u4 & 1) ? m.sPartX : m.sPart, b3v, row, 0);
            if (slices) {
                sQ4[tid] = gk_in + q.gpow[sl] * qv;                                             // mpg_learner.py:266
                sD34[u4 * GROUP * MAXOUT + d3_index(row, 0)] = q.coef[sl];
            } else {    // err = Q(s~,a) - y; dL/dq = err / B_global   (mpg_learner.py:331-336)
                float yv = t_y;
                if (a.qpart) {
                    yv = (t_rew + a.rshift) * a.rscale + a.gamma * fminf(t_q1, t_q2);          // mpg_learner.py:132-133
                    if (qi == 0) a.y_out[gr] = yv;
                }
                const float e = qv - yv + row_poison(sX4 + tid * XS, QIN);
                st.dz3[gr] = e * a.inv_b;
                if (a.td && qi == 0) a.td[gr] = e;
                sD34[u4 * GROUP * MAXOUT + d3_index(row, 0)] = e * a.inv_b;
                sQ4[tid] = e * e;
                report_nan(a.status, e != e);
            }
        }
    }
    report_activation_range(a.status, zmax);
    load_w2<PK>(a.pkb[qi], net.W2, true, L, w2);     // same registers, backward image
    lds_barrier();
    if (tid >= 64 && tid < 64 + G4) {                 // per-group sums (a lane of wave 1 each, fixed order)
        const int u4 = tid - 64;
        float s1 = 0.f, s2 = 0.f;
        for (int i = 0; i < GROUP; ++i) { const float v = sQ4[u4 * GROUP + i]; s1 += v; s2 += v * v; }
        if (slices) {
            q.ret_part[((long)sl * ngroups + g0 + u4) * 2] = s1;
            q.ret_part[((long)sl * ngroups + g0 + u4) * 2 + 1] = s2;
        } else {
            a.loss_part[qi * ngroups + g0 + u4] = 0.5f * a.inv_b * s1;
        }
    }
    // ---- reverse: the four groups as two pairs behind the transposed image (backward_group2): the second pair (registers)
    //      first, then the first pair (back from LDS) ----
    float dz1b[2][4], dz2b[2][4];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int p = 1 - pass, ua = 2 * p, ub = 2 * p + 1;
        if (pass == 1) {
            const f32x4* hold = reinterpret_cast<const f32x4*>(sHold) + tid;
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f32x4 x1 = hold[((e * 2 + 0) * 2 + t) * NTHREAD], x2 = hold[((e * 2 + 1) * 2 + t) * NTHREAD];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { h1[e][t][j] = x1[j]; h2[e][t][j] = x2[j]; }
                }
        }
        if (slices) {
            backward_group2<QIN, 1, true>(sD34 + ua * GROUP * MAXOUT, sD34 + ub * GROUP * MAXOUT, m.sA, sA2, m.sA1, m.sPartX, sPartX2, L, w2, r,
                                          h1[0], h2[0], h1[1], h2[1], dz1, dz2, dz1b, dz2b);
            if (tid < 2 * GROUP * QIN) {
                const int e = tid / (GROUP * QIN), rem = tid % (GROUP * QIN), row = rem / QIN, i = rem % QIN;
                const long gr = ((long)sl * ngroups + g0 + ua + e) * GROUP + row;
                q.gxq[gr * QIN + i] = dx_reduce(e ? sPartX2 : m.sPartX, row, i);
            }
            if (pass == 0) lds_barrier();            // the second pass rewrites the partials these lanes are reading
        } else {
            backward_group2<QIN, 1, false>(sD34 + ua * GROUP * MAXOUT, sD34 + ub * GROUP * MAXOUT, m.sA, sA2, m.sA1, m.sPartX, sPartX2, L, w2, r,
                                           h1[0], h2[0], h1[1], h2[1], dz1, dz2, dz1b, dz2b);
            stash_store(st.dz1, g0 + ua, L, dz1);
            stash_store(st.dz2, g0 + ua, L, dz2);
            stash_store(st.dz1, g0 + ub, L, dz1b);
            stash_store(st.dz2, g0 + ub, L, dz2b);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
struct WgradMulti {
    int n_jobs;
    WgradArgs a[3];
    int type[3];               // 0: template pair A, 1: template pair B
    int chunk_off[4];          // chunk ranges of the jobs
    int chunk0;                // first chunk this launch covers (a launch may cover a suffix of the jobs)
    unsigned long long* dbg;   // MPG_TIMELINE builds only
};

#ifndef MPG_WGM_WAVES
#define MPG_WGM_WAVES 4
#endif
template <int IA, int OA, int IB, int OB>
__global__ void __launch_bounds__(NTHREAD, MPG_WGM_WAVES) k_wgrad_multi(const WgradMulti m) {
    constexpr int NQA = wgrad_nq<IA, OA>(), NQB = wgrad_nq<IB, OB>();
    __shared__ __attribute__((aligned(16))) float sRed[NWAVE * (NQA > NQB ? NQA : NQB) * 64];
    int gchunk, sl;
#if defined(MPG_SPLIT)
    // Two kinds of workgroup (round 4): the first third of the grid takes dW2 in 64-COLUMN slices (four workgroups per chunk re-read
    // the chunk's H1 through L2 instead of eight: mlp_wgrad.h NT = 4), the other two thirds the thin pieces of the (chunk, 32-column
    // slice) pairs - a chain of load round trips that used to run as a tail behind every matrix loop.  12 workgroups per chunk,
    // 576 for the bench step's three jobs on 512 resident slots; the matrix ones are dispatched first.
    const int nch = gridDim.x / 12;
    const int role = (int)blockIdx.x < 4 * nch ? 1 : 2;
    if (role == 1) wgrad_map4(blockIdx.x, nch, gchunk, sl);
    else wgrad_map((int)blockIdx.x - 4 * nch, nch, gchunk, sl);
#else       // every workgroup does both, the thin pieces as a tail behind the matrix loop (the exact-fp32 engine; rounds 2 - 3)
    const int role = 0;
    wgrad_map(blockIdx.x, gridDim.x >> 3, gchunk, sl);
#endif
    gchunk += m.chunk0;
    int j = 0;
    while (j + 1 < m.n_jobs && gchunk >= m.chunk_off[j + 1]) ++j;
    const int chunk = gchunk - m.chunk_off[j];
    MPG_TL_DECL
    MPG_TL(0);
#if defined(MPG_SPLIT)
    if (role == 1) {
        if (m.type[j] == 0) wgrad_body<IA, OA, 1, 4>(m.a[j], sl, chunk, sRed);
        else wgrad_body<IB, OB, 1, 4>(m.a[j], sl, chunk, sRed);
    } else {
        if (m.type[j] == 0) wgrad_body<IA, OA, 2>(m.a[j], sl, chunk, sRed);
        else wgrad_body<IB, OB, 2>(m.a[j], sl, chunk, sRed);
    }
#else
    if (m.type[j] == 0) wgrad_body<IA, OA, 0>(m.a[j], sl, chunk, sRed);
    else wgrad_body<IB, OB, 0>(m.a[j], sl, chunk, sRed);
    (void)role;
#endif
    MPG_TL(7);
#ifdef MPG_TIMELINE
    __syncthreads();
    if (m.dbg && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && threadIdx.x < NWAVE * MPG_TL_MARKS)
        m.dbg[(blockIdx.x ? 1 : 0) * NWAVE * MPG_TL_MARKS + threadIdx.x] = s_tl[threadIdx.x / MPG_TL_MARKS][threadIdx.x % MPG_TL_MARKS];
#endif
}

struct ReduceMulti {
    int n_jobs;
    const float* slabs[3];
    int nslab[3], n[3];
    float* out[3];
    int n_sums;
    SumJob sums[8];
    float* sq_part;        // nullable: [n_jobs][MPG_CLIP_PARTS]
};

// blockIdx.y < n_jobs: out = sum of the chunk slabs (fixed order); blockIdx.y == n_jobs: the scalar sums, one per block
__global__ void __launch_bounds__(256) k_reduce_multi(const ReduceMulti m) {
    __shared__ float red[256];
    const int j = blockIdx.y;
    if (j < m.n_jobs) {
        const int i = blockIdx.x * blockDim.x + threadIdx.x;
        const float* sl = m.slabs[j];
        const int n = m.n[j], ns = m.nslab[j];
        float tot = 0.f;
        if (i < n) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            int k = 0;
            for (; k + 3 < ns; k += 4) {      // 4 loads in flight; the association ((s0+s4+..)+(s1+s5+..))+... is fixed
                acc[0] += sl[(size_t)k * n + i];
                acc[1] += sl[(size_t)(k + 1) * n + i];
                acc[2] += sl[(size_t)(k + 2) * n + i];
                acc[3] += sl[(size_t)(k + 3) * n + i];
            }
            for (; k < ns; ++k) acc[0] += sl[(size_t)k * n + i];
            tot = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            m.out[j][i] = tot;
        }
        if (m.sq_part) {   // the clip's per-block partial sums of squares (k_sq_blocks computes the same numbers)
            const float sq = mpg_block_sum256(fmaf(tot, tot, 0.f), red);
            if (threadIdx.x == 0 && (int)blockIdx.x < MPG_CLIP_PARTS) m.sq_part[j * MPG_CLIP_PARTS + blockIdx.x] = sq;
        }
    } else if ((int)blockIdx.x < m.n_sums) {
        const SumJob sj = m.sums[blockIdx.x];
        float s = 0.f;
        for (int i = threadIdx.x; i < sj.n; i += 256) s += sj.src[(size_t)i * sj.stride];
        const float t = mpg_block_sum256(s, red);
        if (threadIdx.x == 0) sj.dst[0] = t;
    }
}

inline void fill_scale(float (&dst)[8], const mpg_cfg_t* cfg) {
    for (int i = 0; i < 8; ++i) dst[i] = i < cfg->obs_dim ? cfg->obs_scale[i] : 1.f;
}

}  // namespace

int launch_target_fused(const mpg_cfg_t* cfg, const float* policy_t, const float* q1t, const float* q2t, int rows,
                        const float* rew, const float* obs_tp1, const float* smooth_eps, float sigma, float clipc, float* y,
                        hipStream_t s, const mpg_replay_draw_t* draw, const DrawOut* draw_out, float* qpart) {
    const int od = cfg->obs_dim, ad = cfg->act_dim;
    MPG_REQUIRE(!qpart || q2t, "launch_target_fused: the split form needs both target critics");
    TargetArgs a;
    a.qpart = qpart;
    a.draw = 0; a.n_storage = 0; a.dk0 = a.dk1 = a.dc1 = a.dc2 = 0;
    a.d_capacity = a.d_fresh_start = a.d_fresh_count = 0;
    a.r_obs = a.r_act = a.r_rew = a.r_obs2 = nullptr; a.r_done = nullptr; a.o_idx = nullptr;
    a.o_obs = a.o_act = a.o_rew = a.o_obs2 = a.o_done = nullptr;
    if (draw) {
        MPG_REQUIRE(draw_out && draw->n_storage > 0 && draw->ring_obs && draw->ring_act && draw->ring_rew && draw->ring_obs2 &&
                        draw->ring_done && draw_out->obs && draw_out->act && draw_out->rew && draw_out->obs2,
                    "launch_target_fused: incomplete replay draw");
        a.draw = draw->pre_gathered ? 2 : 1; a.n_storage = draw->n_storage;
        a.d_capacity = draw->capacity; a.d_fresh_start = draw->fresh_start; a.d_fresh_count = draw->fresh_count;
        MPG_REQUIRE(!draw->pre_gathered || (draw->capacity > 0 && draw->fresh_start >= 0 && draw->fresh_start < draw->capacity &&
                                            draw->fresh_count >= 0 && draw->fresh_count <= draw->capacity),
                    "launch_target_fused: bad pre-gathered window");
        a.dk0 = (uint32_t)draw->seed; a.dk1 = (uint32_t)(draw->seed >> 32);
        a.dc1 = (uint32_t)draw->ctr; a.dc2 = (uint32_t)(draw->ctr >> 32);
        a.r_obs = draw->ring_obs; a.r_act = draw->ring_act; a.r_rew = draw->ring_rew; a.r_obs2 = draw->ring_obs2;
        a.r_done = draw->ring_done; a.o_idx = draw->idx_out; a.o_done = draw->done_out;
        a.o_obs = draw_out->obs; a.o_act = draw_out->act; a.o_rew = draw_out->rew; a.o_obs2 = draw_out->obs2;
    }
    a.pol = policy_t; a.q1 = q1t; a.q2 = q2t; a.status = mpg_status_of(cfg);
    a.pk_pol = weight_cache_lookup(cfg, make_net(policy_t, od, 2 * ad).W2, 0);
    a.pk_q1 = weight_cache_lookup(cfg, make_net(q1t, od + ad, 1).W2, 0);
    a.pk_q2 = q2t ? weight_cache_lookup(cfg, make_net(q2t, od + ad, 1).W2, 0) : nullptr;
    a.rows = rows; a.obs2 = obs_tp1; a.rew = rew; a.smooth_eps = smooth_eps;
    fill_scale(a.scale, cfg);
    const bool ranged = cfg->action_range > 0.f;
    a.out_tanh = (cfg->policy_out_act == MPG_ACT_TANH || ranged) ? 1 : 0;
    a.out_scale = ranged ? cfg->action_range : 1.f;
    a.sigma = sigma; a.clipc = clipc; a.rshift = cfg->rew_shift; a.rscale = cfg->rew_scale; a.gamma = cfg->gamma; a.y = y;
    const int ngroups = (rows + GROUP - 1) / GROUP;
    a.dbg = nullptr;
#ifdef MPG_TIMELINE
    static unsigned long long* s_dbg = nullptr;
    static int s_calls = 0;
    if (!s_dbg) (void)hipMalloc(&s_dbg, 2 * NWAVE * MPG_TL_MARKS * sizeof(unsigned long long));
    a.dbg = s_dbg;
#endif
    mpg_prof_begin(mpg_prof_of(cfg), 6, s);
    // two row groups per workgroup once there is at least one 16-row group per CU (see k_target_fused)
#ifndef MPG_TARGET_G2_MIN_GROUPS
#define MPG_TARGET_G2_MIN_GROUPS 256
#endif
    const bool pk = a.pk_pol && a.pk_q1 && (a.pk_q2 || !a.q2);
    const bool two = ngroups >= MPG_TARGET_G2_MIN_GROUPS;
    const dim3 grid(two ? (ngroups + 1) / 2 : ngroups, qpart ? 2 : 1), block(NTHREAD);
#define MPG_TARGET_LAUNCH(O_, A_) \
    do { \
        if (pk && two) hipLaunchKernelGGL((k_target_fused<O_, A_, true, 2>), grid, block, 0, s, a); \
        else if (pk) hipLaunchKernelGGL((k_target_fused<O_, A_, true, 1>), grid, block, 0, s, a); \
        else if (two) hipLaunchKernelGGL((k_target_fused<O_, A_, false, 2>), grid, block, 0, s, a); \
        else hipLaunchKernelGGL((k_target_fused<O_, A_, false, 1>), grid, block, 0, s, a); \
    } while (0)
    if (od == 6 && ad == 2) MPG_TARGET_LAUNCH(6, 2);
    else if (od == 4 && ad == 1) MPG_TARGET_LAUNCH(4, 1);
#undef MPG_TARGET_LAUNCH
    else { mpg_set_error("launch_target_fused: unsupported dims"); return MPG_EINVAL; }
    mpg_prof_end(mpg_prof_of(cfg), 6, s);
#ifdef MPG_TIMELINE
    if (++s_calls % 100 == 0) {
        static unsigned long long h[2 * NWAVE * MPG_TL_MARKS];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, s_dbg, sizeof(h), hipMemcpyDeviceToHost);
        for (int b = 0; b < 2; ++b)
            for (int w = 0; w < NWAVE; w += 7) {
                fprintf(stderr, "timeline target wg%d wave%d:", b ? 100 : 0, w);
                const unsigned long long* t = h + (b * NWAVE + w) * MPG_TL_MARKS;
                for (int k = 1; k < MPG_TL_MARKS; ++k) fprintf(stderr, " %d:%lld", k, t[k] ? (long long)(t[k] - t[0]) : -1LL);
                fprintf(stderr, "\n");
            }
    }
#endif
    MPG_CHECK_LAUNCH("k_target_fused");
    return MPG_OK;
}

int launch_qloss_fused(const mpg_cfg_t* cfg, const float* const* q_params, int n_q, int rows, const float* obs,
                       const float* act, const float* y, float inv_b, const CriticStash* st, float* loss_part, float* td,
                       hipStream_t s) {
    MPG_REQUIRE(n_q == 1 || n_q == 2, "launch_qloss_fused: n_q");
    const int od = cfg->obs_dim, ad = cfg->act_dim, qin = od + ad;
    QlossArgs a;
    for (int k = 0; k < 2; ++k) {
        a.q[k] = k < n_q ? q_params[k] : nullptr;
        a.pkf[k] = k < n_q ? weight_cache_lookup(cfg, make_net(q_params[k], qin, 1).W2, 0) : nullptr;
        a.pkb[k] = k < n_q ? weight_cache_lookup(cfg, make_net(q_params[k], qin, 1).W2, 1) : nullptr;
        if (k < n_q) a.st[k] = st[k];
    }
    a.rows = rows; a.x = xspec(obs, od, act, ad, cfg->obs_scale, od); a.y = y; a.inv_b = inv_b; a.loss_part = loss_part; a.td = td;
    a.qpart = a.rew = nullptr; a.y_out = nullptr; a.rshift = a.gamma = 0.f; a.rscale = 1.f;
    a.status = mpg_status_of(cfg);
    const int ngroups = (rows + GROUP - 1) / GROUP;
    if (qin == 8) { if (a.pkf[0] && a.pkb[0] && (n_q < 2 || (a.pkf[1] && a.pkb[1]))) hipLaunchKernelGGL((k_qloss_fused<8, true>), dim3(ngroups, n_q), dim3(NTHREAD), 0, s, a); else hipLaunchKernelGGL((k_qloss_fused<8, false>), dim3(ngroups, n_q), dim3(NTHREAD), 0, s, a); }
    else if (qin == 5) { if (a.pkf[0] && a.pkb[0] && (n_q < 2 || (a.pkf[1] && a.pkb[1]))) hipLaunchKernelGGL((k_qloss_fused<5, true>), dim3(ngroups, n_q), dim3(NTHREAD), 0, s, a); else hipLaunchKernelGGL((k_qloss_fused<5, false>), dim3(ngroups, n_q), dim3(NTHREAD), 0, s, a); }
    else { mpg_set_error("launch_qloss_fused: unsupported dims"); return MPG_EINVAL; }
    MPG_CHECK_LAUNCH("k_qloss_fused");
    return MPG_OK;
}

int launch_qslice_fused(const mpg_cfg_t* cfg, const float* q_params, int qin, int R, int n_sel, const float* xq, const float* gk, const float* gpow,
                        const float* coef, float* ret_part, float* gxq, hipStream_t s) {
    MPG_REQUIRE(R % GROUP == 0 && n_sel >= 1 && n_sel <= 4, "launch_qslice_fused: needs rows %% 16 == 0");
    QsliceArgs a;
    a.q = q_params;
    a.pkf = weight_cache_lookup(cfg, make_net(q_params, qin, 1).W2, 0);
    a.pkb = weight_cache_lookup(cfg, make_net(q_params, qin, 1).W2, 1);
    a.R = R; a.n_sel = n_sel; a.xq = xq; a.gk = gk; a.ret_part = ret_part; a.gxq = gxq; a.status = mpg_status_of(cfg);
    for (int k = 0; k < 4; ++k) { a.gpow[k] = k < n_sel ? gpow[k] : 0.f; a.coef[k] = k < n_sel ? coef[k] : 0.f; }
    const int ngroups = n_sel * (R / GROUP);
    if (n_sel == 2) {
        if (qin == 8) { if (a.pkf && a.pkb) hipLaunchKernelGGL((k_qslice_fused2<8, true>), dim3(R / GROUP), dim3(NTHREAD), 0, s, a); else hipLaunchKernelGGL((k_qslice_fused2<8, false>), dim3(R / GROUP), dim3(NTHREAD), 0, s, a); }
        else if (qin == 5) { if (a.pkf && a.pkb) hipLaunchKernelGGL((k_qslice_fused2<5, true>), dim3(R / GROUP), dim3(NTHREAD), 0, s, a); else hipLaunchKernelGGL((k_qslice_fused2<5, false>), dim3(R / GROUP), dim3(NTHREAD), 0, s, a); }
        else { mpg_set_error("launch_qslice_fused: unsupported dims"); return MPG_EINVAL; }
    } else if (qin == 8) { if (a.pkf && a.pkb) hipLaunchKernelGGL((k_qslice_fused<8, true>), dim3(ngroups), dim3(NTHREAD), 0, s, a); else hipLaunchKernelGGL((k_qslice_fused<8, false>), dim3(ngroups), dim3(NTHREAD), 0, s, a); }
    else if (qin == 5) { if (a.pkf && a.pkb) hipLaunchKernelGGL((k_qslice_fused<5, true>), dim3(ngroups), dim3(NTHREAD), 0, s, a); else hipLaunchKernelGGL((k_qslice_fused<5, false>), dim3(ngroups), dim3(NTHREAD), 0, s, a); }
    else { mpg_set_error("launch_qslice_fused: unsupported dims"); return MPG_EINVAL; }
    MPG_CHECK_LAUNCH("k_qslice_fused");
    return MPG_OK;
}

int launch_critic_fused(const mpg_cfg_t* cfg, const float* const* q_params, int n_q, int rows, const float* obs,
                        const float* act, const float* y, float inv_b, const CriticStash* st, float* loss_part, const float* xq,
                        const float* gk, const float* gpow, const float* coef, float* ret_part, float* gxq, hipStream_t s,
                        const float* qpart, const float* rew, float* y_out) {
    MPG_REQUIRE((n_q == 1 || n_q == 2) && rows % GROUP == 0, "launch_critic_fused: n_q / rows");
    MPG_REQUIRE(!qpart || (rew && y_out), "launch_critic_fused: the split target needs rew and y_out");
    const int od = cfg->obs_dim, ad = cfg->act_dim, qin = od + ad;
    CriticArgs c;
    QlossArgs& a = c.ql;
    for (int k = 0; k < 2; ++k) {
        a.q[k] = k < n_q ? q_params[k] : nullptr;
        a.pkf[k] = k < n_q ? weight_cache_lookup(cfg, make_net(q_params[k], qin, 1).W2, 0) : nullptr;
        a.pkb[k] = k < n_q ? weight_cache_lookup(cfg, make_net(q_params[k], qin, 1).W2, 1) : nullptr;
        if (k < n_q) a.st[k] = st[k];
    }
    a.rows = rows; a.x = xspec(obs, od, act, ad, cfg->obs_scale, od); a.y = y; a.inv_b = inv_b; a.loss_part = loss_part; a.td = nullptr;
    a.qpart = qpart; a.rew = rew; a.y_out = y_out; a.rshift = cfg->rew_shift; a.rscale = cfg->rew_scale; a.gamma = cfg->gamma;
    a.status = mpg_status_of(cfg);
    QsliceArgs& q = c.qs;
    q.q = q_params[0]; q.pkf = a.pkf[0]; q.pkb = a.pkb[0];
    q.R = rows; q.n_sel = 2; q.xq = xq; q.gk = gk; q.ret_part = ret_part; q.gxq = gxq; q.status = a.status;
    for (int k = 0; k < 4; ++k) { q.gpow[k] = k < 2 ? gpow[k] : 0.f; q.coef[k] = k < 2 ? coef[k] : 0.f; }
    const int ngroups = rows / GROUP;
    c.dbg = nullptr;
#ifdef MPG_TIMELINE
    static unsigned long long* s_dbg = nullptr;
    static int s_calls = 0;
    if (!s_dbg) (void)hipMalloc(&s_dbg, 2 * NWAVE * MPG_TL_MARKS * sizeof(unsigned long long));
    c.dbg = s_dbg;
#endif
    mpg_prof_begin(mpg_prof_of(cfg), 7, s);
    // balanced form (k_critic_fused4): four groups of one kind per workgroup, two image loads per CU instead of four - once there is
    // at least one workgroup per CU that way and the groups divide evenly
#ifndef MPG_CRITIC4_MIN_GROUPS
#define MPG_CRITIC4_MIN_GROUPS 256
#endif
    const bool packed = a.pkf[0] && a.pkb[0] && (n_q < 2 || (a.pkf[1] && a.pkb[1]));
    if (packed && ngroups >= MPG_CRITIC4_MIN_GROUPS && ngroups % 4 == 0 && (qin == 8 || qin == 5)) {
        const int nwg = (3 + (n_q == 2 ? 1 : 0)) * (ngroups / 4);
        if (qin == 8) hipLaunchKernelGGL((k_critic_fused4<8, true>), dim3(nwg), dim3(NTHREAD), 0, s, c);
        else hipLaunchKernelGGL((k_critic_fused4<5, true>), dim3(nwg), dim3(NTHREAD), 0, s, c);
    } else
    if (qin == 8) { if (a.pkf[0] && a.pkb[0] && (n_q < 2 || (a.pkf[1] && a.pkb[1]))) hipLaunchKernelGGL((k_critic_fused<8, true>), dim3(ngroups, n_q), dim3(NTHREAD), 0, s, c); else hipLaunchKernelGGL((k_critic_fused<8, false>), dim3(ngroups, n_q), dim3(NTHREAD), 0, s, c); }
    else if (qin == 5) { if (a.pkf[0] && a.pkb[0] && (n_q < 2 || (a.pkf[1] && a.pkb[1]))) hipLaunchKernelGGL((k_critic_fused<5, true>), dim3(ngroups, n_q), dim3(NTHREAD), 0, s, c); else hipLaunchKernelGGL((k_critic_fused<5, false>), dim3(ngroups, n_q), dim3(NTHREAD), 0, s, c); }
    else { mpg_set_error("launch_critic_fused: unsupported dims"); return MPG_EINVAL; }
    mpg_prof_end(mpg_prof_of(cfg), 7, s);
#ifdef MPG_TIMELINE
    if (++s_calls % 100 == 0) {
        static unsigned long long h[2 * NWAVE * MPG_TL_MARKS];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, s_dbg, sizeof(h), hipMemcpyDeviceToHost);
        for (int b = 0; b < 2; ++b)
            for (int w = 0; w < NWAVE; w += 7) {
                fprintf(stderr, "timeline critic Q%d wg7 wave%d:", b + 1, w);
                const unsigned long long* t = h + (b * NWAVE + w) * MPG_TL_MARKS;
                for (int k = 1; k < 8; ++k) fprintf(stderr, " %d:%lld", k, (long long)(t[k] - t[0]));
                fprintf(stderr, "\n");
            }
    }
#endif
    MPG_CHECK_LAUNCH("k_critic_fused");
    return MPG_OK;
}

// phases: 1 = the chunk products of jobs[first_job .. n_jobs) only, 2 = the slab reduction (+ sums) of jobs[0 .. n_jobs) only,
// 3 = both (one weight-gradient launch over all jobs, then the reduction)
int launch_wgrad_multi(const mpg_cfg_t* cfg, const WgradJob* jobs, int n_jobs, const SumJob* sums, int n_sums, float* sq_part, hipStream_t s,
                       int phases, int first_job) {
    MPG_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= 3 && n_sums >= 0 && n_sums <= 8 && first_job >= 0 && first_job < n_jobs, "launch_wgrad_multi: bad argument");
    WgradMulti m;
    ReduceMulti rm;
    m.n_jobs = rm.n_jobs = n_jobs;
    int off = 0, maxn = 0;
    bool pendulum = false;
    for (int j = 0; j < n_jobs; ++j) {
        const WgradJob& jb = jobs[j];
        WgradArgs& a = m.a[j];
        a.no_thin = 0; a.in_dim = jb.in_dim; a.out_dim = jb.out_dim; a.rows = jb.rows; a.x = jb.x;
        a.h1 = jb.h1; a.h2 = jb.h2; a.dz1 = jb.dz1; a.dz2 = jb.dz2; a.dz3 = jb.dz3; a.slabs = jb.slabs;
        const long ngroups = (jb.rows + GROUP - 1) / GROUP;
        a.groups_per_chunk = wgrad_groups_per_chunk(ngroups);
        const int nch = (int)((ngroups + a.groups_per_chunk - 1) / a.groups_per_chunk);
        m.chunk_off[j] = off;
        off += nch;
        if (jb.in_dim == 8 && jb.ou == 1) m.type[j] = 0;
        else if (jb.in_dim == 6 && jb.ou == 2) m.type[j] = 1;
        else if (jb.in_dim == 5 && jb.ou == 1) { m.type[j] = 0; pendulum = true; }
        else if (jb.in_dim == 4 && jb.ou == 1) { m.type[j] = 1; pendulum = true; }
        else { mpg_set_error("launch_wgrad_multi: unsupported network shape"); return MPG_EINVAL; }
        rm.slabs[j] = jb.slabs; rm.nslab[j] = nch; rm.n[j] = net_size(jb.in_dim, jb.out_dim); rm.out[j] = jb.grad;
        if (rm.n[j] > maxn) maxn = rm.n[j];
    }
    m.chunk_off[n_jobs] = off;
    const int chunk0 = m.chunk_off[first_job];          // the launch covers the chunks of jobs[first_job ..) only
    for (int j = n_jobs; j < 3; ++j) { m.type[j] = 0; m.chunk_off[j + 1] = off; rm.slabs[j] = nullptr; rm.nslab[j] = rm.n[j] = 0; rm.out[j] = nullptr; }
    m.dbg = nullptr;
#ifdef MPG_TIMELINE
    static unsigned long long* s_dbg = nullptr;
    static int s_calls = 0;
    if (!s_dbg) (void)hipMalloc(&s_dbg, 2 * NWAVE * MPG_TL_MARKS * sizeof(unsigned long long));
    m.dbg = s_dbg;
#endif
    m.chunk0 = chunk0;
    if (phases & 1) {
    mpg_prof_begin(mpg_prof_of(cfg), 5, s);
#if defined(MPG_SPLIT)
    const int wg_per_chunk = 12;          // 4 matrix (64-column slices) + 8 thin (32-column slices)
#else
    const int wg_per_chunk = 8;
#endif
    if (!pendulum) hipLaunchKernelGGL((k_wgrad_multi<8, 1, 6, 2>), dim3(wg_per_chunk * (off - chunk0)), dim3(NTHREAD), 0, s, m);
    else hipLaunchKernelGGL((k_wgrad_multi<5, 1, 4, 1>), dim3(wg_per_chunk * (off - chunk0)), dim3(NTHREAD), 0, s, m);
    mpg_prof_end(mpg_prof_of(cfg), 5, s);
    }
#ifdef MPG_TIMELINE
    if (++s_calls % 100 == 0) {
        static unsigned long long h[2 * NWAVE * MPG_TL_MARKS];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, s_dbg, sizeof(h), hipMemcpyDeviceToHost);
        for (int b = 0; b < 2; ++b)
            for (int w = 0; w < NWAVE; w += 7) {
                fprintf(stderr, "timeline wgrad %s wave%d:", b ? "last-wg(policy)" : "wg0(Q1)", w);
                const unsigned long long* t = h + (b * NWAVE + w) * MPG_TL_MARKS;
                for (int k = 1; k < 8; ++k) fprintf(stderr, " %d:%lld", k, (long long)(t[k] - t[0]));
                fprintf(stderr, "\n");
            }
    }
#endif
    MPG_CHECK_LAUNCH("k_wgrad_multi");
    if (!(phases & 2)) return MPG_OK;
    rm.n_sums = n_sums;
    for (int k = 0; k < 8; ++k) {
        if (k < n_sums) rm.sums[k] = sums[k];
        else { rm.sums[k].src = nullptr; rm.sums[k].n = 0; rm.sums[k].stride = 1; rm.sums[k].dst = nullptr; }
    }
    int gx = (maxn + 255) / 256;
    if (gx < n_sums) gx = n_sums;
    // the partials cover the grid only if every network fits MPG_CLIP_PARTS blocks and unused slots are written (as 0)
    rm.sq_part = nullptr;
    if (sq_part) {
        MPG_REQUIRE(gx <= MPG_CLIP_PARTS, "launch_wgrad_multi: network larger than MPG_CLIP_PARTS * 256");
        rm.sq_part = sq_part;
        gx = MPG_CLIP_PARTS;
    }
    hipLaunchKernelGGL(k_reduce_multi, dim3(gx, n_jobs + (n_sums ? 1 : 0)), dim3(256), 0, s, rm);
    MPG_CHECK_LAUNCH("k_reduce_multi");
    return MPG_OK;
}

}  // namespace mlp
