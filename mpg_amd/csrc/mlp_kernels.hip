// K3/K5: generic network kernels on the weight-stationary engine of mlp_core.h:
//   forward (optionally stashing hidden activations), input-side backward, weight gradient (+ slab reduce).
#include "mlp_wgrad.h"

namespace mlp {

// -------------------------------------------------------------------------------------------------------
// forward
// -------------------------------------------------------------------------------------------------------
struct FwdArgs {
    const float* params;
    int in_dim, out_dim, rows;
    XSpec x;
    int out_tanh;
    float out_scale, sigma;
    uint32_t k0, k1, c1, c2;
    float* y;
    int ldy;
    float *h1, *h2;
    const float* pack;   // nullable: packed forward image of W2 (weight cache)
    int* status;         // nullable: MPG_STATUS_* word of the caller
};

// G2 = 2: a workgroup takes its row groups in PAIRS through one pair of barriers (forward_group2): with many groups per
// workgroup (TD3 at B = 65 536: 16) a group's pass is ~4.9k cycles of which the matrix block is 0.8k; the second group of a
// pair rides in the first one's barrier waits and LDS round trips.  Same arithmetic per group (bit-identical results).
template <int IN, int OU, bool PK, int G2>
__global__ void __launch_bounds__(NTHREAD, 2) k_forward(const FwdArgs a) {
    constexpr int XSW = xs_of<IN>();
    __shared__ __attribute__((aligned(16))) float smem[G2 * (A_IMG + GROUP * XSW + NWAVE * GROUP * MAXOUT)];
    float* sA = smem;
    float* sX = sA + G2 * A_IMG;
    float* sPart = sX + G2 * GROUP * XSW;
    const Lane L;
    const Net net = make_net(a.params, a.in_dim, a.out_dim);
    float w2[128];
    SmallRegs<IN, OU> r;
    const long ngroups = (a.rows + GROUP - 1) / GROUP;
    const long nunits = (ngroups + G2 - 1) / G2;                // units of G2 consecutive groups
    // the first unit's inputs are requested BEFORE the small pieces and the 256 KB image and consumed after them: loads
    // retire in order, so layer 1 and its barrier then run while the image is still streaming in
    // Several units per workgroup (one workgroup per CU: nothing else covers a load's way to memory and back): the NEXT unit's inputs
    // are requested at the top of a unit, stay in flight across its passes and go into the other half of a double-buffered input
    // block BEFORE the unit's stash stores are issued - the vector-memory counter retires in order and the wait-count bookkeeping
    // drains it at a loop-carried value, so a register carried across the back edge would wait for those stores.  RAW values are
    // requested: a thread's input column, and with it its scale, is the same for every unit; the product is taken on arrival.
    __shared__ __attribute__((aligned(16))) float sX2[2][G2 * GROUP * XSW];
    float xsc = 1.f;
    if (threadIdx.x < G2 * GROUP * XSW) {
        const int i = threadIdx.x % XSW;
        if (i < a.in_dim && i < a.x.d0) xsc = a.x.scale[i];
    }
    auto x_value = [&](long u) {
        float v = 0.f;
        if (threadIdx.x < G2 * GROUP * XSW) {
            const int g2 = threadIdx.x / (GROUP * XSW), e = threadIdx.x % (GROUP * XSW), row = e / XSW, i = e % XSW;
            const long gr = (u * G2 + g2) * GROUP + row;
            if (gr < a.rows && i < a.in_dim) v = i < a.x.d0 ? a.x.x0[gr * a.x.ld0 + i] : a.x.x1[gr * a.x.ld1 + (i - a.x.d0)];
        }
        return v;
    };
    float xv = x_value(blockIdx.x);
    int cur = 0;
    float b3v = 0.f;             // this output thread's bias, requested up front (at its point of use it is a memory round trip)
    if (threadIdx.x < G2 * GROUP * OU) b3v = net.b3[threadIdx.x % OU];
    float zmax = 0.f;
    bool saw_nan = false;                       // worker.py:95-107 judge_is_nan, on the device: inputs and outputs of the pass
    load_small<IN, OU>(net, L, r);
    if constexpr (PK) load_w2_packed(a.pack, L, w2); else load_w2_fwd(net.W2, L, w2);
    xv *= xsc;                                   // (x * 1.f is x: unscaled columns keep their bits)
    saw_nan |= xv != xv;
    if (threadIdx.x < G2 * GROUP * XSW) sX2[0][threadIdx.x] = xv;
    // the image is consumed ONCE here: left pending across the loop header, the wait-count bookkeeping drains the counter (vmcnt(0)) in
    // every unit's matrix block - and with it the input request issued just before.  (What a first unit loses - its layer 1 under the
    // image request - measured nothing: EXPERIMENTS.md section 5.7.)
#pragma unroll
    for (int v = 0; v < 32; ++v) asm volatile("" ::"v"(w2[4 * v]), "v"(w2[4 * v + 1]), "v"(w2[4 * v + 2]), "v"(w2[4 * v + 3]));
    for (long u = blockIdx.x; u < nunits; u += gridDim.x) {
        lds_barrier();
        sX = sX2[cur];
        xv = u + gridDim.x < nunits ? x_value(u + gridDim.x) : 0.f;
        // a NaN among a row's inputs stays a NaN in its outputs (row_poison, mlp_core.h): read here, while the block is this
        // unit's - the next unit's inputs are stored in front of the barrier above
        float pz = 0.f;
        if (threadIdx.x < G2 * GROUP * OU) pz = row_poison(sX + (threadIdx.x / OU) * XSW, XSW);
        float h1[G2][2][4], h2[G2][2][4];
        if constexpr (G2 == 2)
            forward_group2<IN, OU>(sX, sX + GROUP * XSW, sA, sA + A_IMG, sPart, sPart + NWAVE * GROUP * MAXOUT, L, w2, r, h1[0], h2[0], h1[1],
                                   h2[1], &zmax);
        else
            forward_group<IN, OU>(sX, sA, sPart, L, w2, r, h1[0], h2[0], nullptr, 0, nullptr, &zmax);
        // the next unit's inputs, ahead of this unit's stores (ordered by the barrier at the top).  Unconditional - behind the last
        // unit it stores a zero nobody reads: as a second `if (more)` the consumption could be skipped on a path on which the request
        // was made, as far as the wait-count bookkeeping can tell, and the counter was drained at the top of every unit
        xv *= xsc;
        saw_nan |= xv != xv;
        if (threadIdx.x < G2 * GROUP * XSW) sX2[cur ^ 1][threadIdx.x] = xv;
        cur ^= 1;
#pragma unroll
        for (int g2 = 0; g2 < G2; ++g2) {
            const long g = u * G2 + g2;
            if (g < ngroups) {
                if (a.h1) stash_store(a.h1, g, L, h1[g2]);
                if (a.h2) stash_store(a.h2, g, L, h2[g2]);
            }
        }
        const int tid = threadIdx.x;
        if (tid < G2 * GROUP * OU) {
            const int g2 = tid / (GROUP * OU), row = (tid / OU) % GROUP, o = tid % OU;
            const long gr = (u * G2 + g2) * GROUP + row;
            if (gr < a.rows) {
                float z = out_preact(sPart + g2 * NWAVE * GROUP * MAXOUT, b3v, row, o);
                float y = a.out_tanh ? a.out_scale * tanhf(z) : z;
                y += pz;
                if (a.sigma > 0.f) {   // OffPolicyWorker.sample: action += N(0, sigma), worker.py:97-98
                    Philox4 p = philox4x32_10((uint32_t)gr, a.c1, a.c2, 0x5eedu + (uint32_t)o, a.k0, a.k1);
                    float u1 = u01(p.v[0]), u2 = u01(p.v[1]);
                    y = add_gauss_noise(y, a.sigma, u1, u2);
                }
                saw_nan |= y != y;
                a.y[gr * a.ldy + o] = y;
            }
        }
        // (sX / sPart of the next unit are ordered behind these reads by the barrier at the top of the loop)
    }
    report_activation_range(a.status, zmax);
    if (a.status && saw_nan) atomicOr(a.status, MPG_STATUS_NAN);
}

#define MPG_DISPATCH_NET(in_dim, ou, CALL)                                    \
    if ((in_dim) == 6 && (ou) == 2) { CALL(6, 2); }                           \
    else if ((in_dim) == 8 && (ou) == 1) { CALL(8, 1); }                      \
    else if ((in_dim) == 4 && (ou) == 1) { CALL(4, 1); }                      \
    else if ((in_dim) == 5 && (ou) == 1) { CALL(5, 1); }                      \
    else if ((in_dim) == 6 && (ou) == 1) { CALL(6, 1); }                      \
    else if ((in_dim) > 8 && (in_dim) <= 16 && (ou) == 2) { CALL(16, 2); }    \
    else if ((in_dim) > 8 && (in_dim) <= 16 && (ou) == 1) { CALL(16, 1); }    \
    else if ((in_dim) == 7 && (ou) == 2) { CALL(16, 2); }                     \
    else if ((in_dim) == 8 && (ou) == 2) { CALL(16, 2); }                     \
    else if ((in_dim) > 16 && (in_dim) <= 24 && (ou) == 1) { CALL(24, 1); }   \
    else {                                                                    \
        mpg_set_error("unsupported network shape in=%d used-out=%d", (in_dim), (ou)); \
        return MPG_EINVAL;                                                    \
    }

static int grid_for(long ngroups) { return (int)(ngroups < 256 ? ngroups : 256); }

int launch_forward(const mpg_cfg_t* cfg, const float* params, int in_dim, int out_dim, int ou, int rows, const XSpec& x,
                   const OutSpec& o, float* y, int ldy, float* h1, float* h2, hipStream_t s) {
    MPG_REQUIRE(params && rows > 0 && y && x.d0 + x.d1 == in_dim && ou <= out_dim, "launch_forward: bad argument");
    FwdArgs a;
    a.params = params; a.in_dim = in_dim; a.out_dim = out_dim; a.rows = rows; a.x = x;
    a.out_tanh = o.out_tanh; a.out_scale = o.out_scale; a.sigma = o.sigma;
    a.k0 = (uint32_t)o.seed; a.k1 = (uint32_t)(o.seed >> 32); a.c1 = (uint32_t)o.ctr; a.c2 = (uint32_t)(o.ctr >> 32);
    a.y = y; a.ldy = ldy; a.h1 = h1; a.h2 = h2;
    a.pack = weight_cache_lookup(cfg, make_net(params, in_dim, out_dim).W2, 0);
    a.status = mpg_status_of(cfg);
    const long ngroups = (rows + GROUP - 1) / GROUP;
    mpg_prof_begin(mpg_prof_of(cfg), 3, s);
    // pairs of groups per pass once a workgroup has at least two groups to take
#ifndef MPG_FORWARD_G2_MIN_GROUPS
#define MPG_FORWARD_G2_MIN_GROUPS 512
#endif
    const bool two = ngroups >= MPG_FORWARD_G2_MIN_GROUPS;
    const int grid = grid_for(two ? (ngroups + 1) / 2 : ngroups);
#define CALL(I, O)                                                                                                  \
    if (a.pack && two) hipLaunchKernelGGL((k_forward<I, O, true, 2>), dim3(grid), dim3(NTHREAD), 0, s, a);           \
    else if (a.pack) hipLaunchKernelGGL((k_forward<I, O, true, 1>), dim3(grid), dim3(NTHREAD), 0, s, a);             \
    else if (two) hipLaunchKernelGGL((k_forward<I, O, false, 2>), dim3(grid), dim3(NTHREAD), 0, s, a);               \
    else hipLaunchKernelGGL((k_forward<I, O, false, 1>), dim3(grid), dim3(NTHREAD), 0, s, a)
    MPG_DISPATCH_NET(in_dim, ou, CALL)
#undef CALL
    mpg_prof_end(mpg_prof_of(cfg), 3, s);
    MPG_CHECK_LAUNCH("k_forward");
    return MPG_OK;
}

// -------------------------------------------------------------------------------------------------------
// input-side backward
// -------------------------------------------------------------------------------------------------------
struct BwdArgs {
    const float* params;
    int in_dim, out_dim, rows;
    const float* dy;
    int lddy;
    const float* yout;
    int ldyo, out_tanh;
    float out_scale;
    const float *h1, *h2;
    float *dz1, *dz2, *dz3, *dx;
    int lddx;
    const float* pack;   // nullable: packed backward image of W2
    XSpec x;             // THIN instantiations only: the network input (for dW1)
    float* thin_part;    // THIN instantiations only: [gridDim.x][thin_floats(in_dim, out_dim)] per-workgroup sums
};

// THIN: the thin parameter gradients (dW1, db1, db2, dW3, db3) are accumulated here as per-lane running sums over the workgroup's
// row groups - everything they are made of (x, h2, dz3, dz1, dz2) is in registers or LDS at this point - and written as ONE partial
// per workgroup (mlp_launch.h thin_floats).  The dz1 stash is not written, the weight-gradient launch reads h1 and dz2 only
// (launch_wgrad no_thin) and launch_thin_reduce adds the partials: at 65 536 rows that is 67 MB less written here and 134 MB less
// read there per network (TD3, B = 65 536: k_wgrad 79 -> see DESIGN 4.3).
template <int IN, int OU, bool WANT_DX, bool PK, bool THIN = false>
__global__ void __launch_bounds__(NTHREAD, 2) k_backward(const BwdArgs a) {
    static_assert(!THIN || (!WANT_DX && IN <= 8), "THIN is built for the base input widths, without dx");
    constexpr int XSW = xs_of<IN>();
    __shared__ __attribute__((aligned(16))) float smem[2 * A_IMG + GROUP * MAXOUT + NWAVE * GROUP * XSW];
    __shared__ __attribute__((aligned(16))) float sXt[THIN ? 2 * GROUP * XSW : 4];      // network input of the group, two parities
    float* sA = smem;
    float* sA1 = sA + A_IMG;
    float* sD3 = sA1 + A_IMG;
    float* sPartX = sD3 + GROUP * MAXOUT;
    const Lane L;
    const Net net = make_net(a.params, a.in_dim, a.out_dim);
    float w2t[128];
    SmallRegs<IN, OU> r;
    if constexpr (PK) load_w2_packed(a.pack, L, w2t); else load_w2_bwd(net.W2, L, w2t);
    load_small<IN, OU>(net, L, r);
    const long ngroups = (a.rows + GROUP - 1) / GROUP;
    const int tid = threadIdx.x;
    // THIN: running sums of this lane's two columns over its four rows of every group (rows beyond the batch carry dz = 0)
    float gb1[2] = {0.f, 0.f}, gb2[2] = {0.f, 0.f}, gw3[2][OU], gw1[2][IN], gb3 = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int k = 0; k < OU; ++k) gw3[t][k] = 0.f;
#pragma unroll
        for (int i = 0; i < IN; ++i) gw1[t][i] = 0.f;
    }
    int par = 0;
    // Software pipeline over the row groups (one workgroup per CU: nothing else covers a load's way to memory and back): the loads of
    // group g + 1 - two stash fragments each of h1 and h2, the output gradient, the THIN input row - are requested at the top of
    // group g, stay in flight across its pass and are consumed behind it, AHEAD of the group's own stash stores (the vector-memory
    // counter retires in order, and the wait-count bookkeeping drains it at a loop-carried value).  Requested at the top of their own
    // group they were a memory round trip in front of every group's first barrier: 16 per launch at B = 65 536.
    struct Req {
        float h1[2][4], h2[2][4], dy, yo, x;
    };
    float xsc = 1.f;
    if constexpr (THIN) {
        if (tid < GROUP * XSW && tid % XSW < a.x.d0) xsc = a.x.scale[tid % XSW];
    }
    auto request = [&](long g, Req& q) {
        stash_load(a.h1, g, L, q.h1);
        stash_load(a.h2, g, L, q.h2);
        q.dy = q.yo = q.x = 0.f;
        if (tid < GROUP * OU) {
            const long gr = g * GROUP + tid / OU;
            if (gr < a.rows) {
                q.dy = a.dy[gr * a.lddy + tid % OU];
                if (a.out_tanh) q.yo = a.yout[gr * a.ldyo + tid % OU];
            }
        }
        if constexpr (THIN) {
            if (tid < GROUP * XSW) {
                const int i = tid % XSW;
                const long gr = g * GROUP + tid / XSW;
                if (gr < a.rows && i < a.x.d0 + a.x.d1) q.x = i < a.x.d0 ? a.x.x0[gr * a.x.ld0 + i] : a.x.x1[gr * a.x.ld1 + (i - a.x.d0)];
            }
        }
    };
    // the output gradient of group g into sD3 (and the caller's dz3), the THIN input row into the parity's block
    auto publish = [&](long g, const Req& q, int parity) {
        if (tid < GROUP * OU) {
            const int row = tid / OU, o = tid % OU;
            const long gr = g * GROUP + row;
            float d = 0.f;
            if (gr < a.rows) {
                d = q.dy;
                if (a.out_tanh) {   // a = S*tanh(z): da/dz = S*(1 - (a/S)^2)
                    const float t = q.yo / a.out_scale;
                    d *= a.out_scale * (1.f - t * t);
                }
                if (a.dz3) a.dz3[gr * OU + o] = d;
            }
            sD3[d3_index(row, o)] = d;
            if constexpr (THIN) gb3 += d;
        }
        if constexpr (THIN) {
            if (tid < GROUP * XSW) sXt[parity * GROUP * XSW + tid] = q.x * xsc;        // (x * 1.f is x)
        }
    };
    Req nx;
    float h1[2][4], h2[2][4], dz1[2][4], dz2[2][4];
    if ((long)blockIdx.x < ngroups) {
        request(blockIdx.x, nx);
        publish(blockIdx.x, nx, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) { h1[t][j] = nx.h1[t][j]; h2[t][j] = nx.h2[t][j]; }
    }
    // (the transposed image is consumed once, ahead of the loop: see k_forward)
#pragma unroll
    for (int v = 0; v < 32; ++v) asm volatile("" ::"v"(w2t[4 * v]), "v"(w2t[4 * v + 1]), "v"(w2t[4 * v + 2]), "v"(w2t[4 * v + 3]));
    for (long g = blockIdx.x; g < ngroups; g += gridDim.x, par ^= 1) {
        lds_barrier();                           // sD3 / the THIN input block of this group are in place
        const long gn = g + gridDim.x < ngroups ? g + gridDim.x : g;      // (behind the last group: its own, again - nobody reads it)
        request(gn, nx);
        if constexpr (THIN) {           // dW3 += h2 dz3^T (sD3 is stable until backward_group's last barrier)
#pragma unroll
            for (int k = 0; k < OU; ++k) {
                const f32x4 d3 = *reinterpret_cast<const f32x4*>(sD3 + d3_index(4 * L.rg, k));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    gw3[0][k] = fmaf(h2[0][j], d3[j], gw3[0][k]);
                    gw3[1][k] = fmaf(h2[1][j], d3[j], gw3[1][k]);
                }
            }
        }
        backward_group<IN, OU, WANT_DX>(sD3, sA, sA1, sPartX, L, w2t, r, h1, h2, dz1, dz2);
        // the next group's requests, consumed ahead of this group's stores (sD3 was last read in front of backward_group's final
        // barrier; the input block of the other parity two groups ago)
        if (g + gridDim.x < ngroups) publish(g + gridDim.x, nx, par ^ 1);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) { h1[t][j] = nx.h1[t][j]; h2[t][j] = nx.h2[t][j]; }
        if (!THIN && a.dz1) stash_store(a.dz1, g, L, dz1);
        if (a.dz2) stash_store(a.dz2, g, L, dz2);
        if constexpr (THIN) {           // db2 += dz2, db1 += dz1, dW1 += x^T dz1
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gb2[0] += dz2[0][j]; gb2[1] += dz2[1][j];
                gb1[0] += dz1[0][j]; gb1[1] += dz1[1][j];
                const float* xr = sXt + par * GROUP * XSW + L.row(j) * XSW;
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(xr), x1 = *reinterpret_cast<const f32x4*>(xr + 4);
#pragma unroll
                for (int i = 0; i < IN; ++i) {
                    const float xv = i < 4 ? x0[i & 3] : x1[i & 3];
                    gw1[0][i] = fmaf(xv, dz1[0][j], gw1[0][i]);
                    gw1[1][i] = fmaf(xv, dz1[1][j], gw1[1][i]);
                }
            }
        }
        if (WANT_DX) {
            if (tid < GROUP * a.in_dim) {
                const int row = tid / a.in_dim, i = tid % a.in_dim;
                const long gr = g * GROUP + row;
                if (gr < a.rows) a.dx[gr * a.lddx + i] = dx_reduce<XSW>(sPartX, row, i);
            }
            lds_barrier();   // sPartX / sD3 are rewritten by the next group
        }
    }
    if constexpr (THIN) {
        // this workgroup's partial: the four row quads (lanes c, c + 16, c + 32, c + 48) of every column summed in a fixed order,
        // written by the quad-0 lane in the network's flat layout without W2
        auto quad_sum = [](float v) {
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            return v;
        };
        float* dst = a.thin_part + (size_t)blockIdx.x * thin_floats(a.in_dim, a.out_dim);
        float *dW1 = dst, *db1 = dW1 + a.in_dim * H, *db2 = db1 + H, *dW3 = db2 + H, *db3 = dW3 + H * a.out_dim;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int col = L.col(t);
            const float s1 = quad_sum(gb1[t]), s2 = quad_sum(gb2[t]);
            if (L.rg == 0) { db1[col] = s1; db2[col] = s2; }
#pragma unroll
            for (int k = 0; k < OU; ++k) {
                const float s3 = quad_sum(gw3[t][k]);
                if (L.rg == 0) dW3[col * a.out_dim + k] = s3;
            }
            if (L.rg == 0)
                for (int k = OU; k < a.out_dim; ++k) dW3[col * a.out_dim + k] = 0.f;     // unused outputs (SURVEY B-5)
#pragma unroll
            for (int i = 0; i < IN; ++i) {
                const float sw = quad_sum(gw1[t][i]);
                if (L.rg == 0) dW1[i * H + col] = sw;
            }
        }
        __syncthreads();                                     // every wave is done with sD3
        if (tid < GROUP * OU) sD3[tid] = gb3;                // [row][o]
        __syncthreads();
        if (tid < a.out_dim) {
            float s3 = 0.f;
            if (tid < OU)
                for (int row = 0; row < GROUP; ++row) s3 += sD3[row * OU + tid];
            db3[tid] = s3;
        }
    }
}

// out[i (+ skip behind skip_from)] = sum over n_part partial arrays of n floats each, in a FIXED association (deterministic): SL lanes per
// output take the parts sl, sl + SL, ... as four running sums each, then the SL slice sums are added in slice order.  (One lane per
// output walking 64 - 256 parts is a chain of dependent-latency steps: 20 - 30 us per launch at 256 parts, measured in the TD3 step.)
template <int SL>
__global__ void __launch_bounds__(256) k_sum_parts(const float* __restrict__ part, int n_part, int n, int skip_from, int skip,
                                                   float* __restrict__ out) {
    constexpr int NO = 256 / SL;                       // outputs per block
    __shared__ float sR[SL][NO + 1];
    const int o = threadIdx.x % NO, sl = threadIdx.x / NO;
    const int i = blockIdx.x * NO + o;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < n) {
        int k = sl;
        for (; k + 3 * SL < n_part; k += 4 * SL) {
            acc[0] += part[(size_t)k * n + i];
            acc[1] += part[(size_t)(k + SL) * n + i];
            acc[2] += part[(size_t)(k + 2 * SL) * n + i];
            acc[3] += part[(size_t)(k + 3 * SL) * n + i];
        }
        for (; k < n_part; k += SL) acc[0] += part[(size_t)k * n + i];
    }
    sR[sl][o] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (sl == 0 && i < n) {
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < SL; ++q) sum += sR[q][o];
        out[i < skip_from ? i : i + skip] = sum;
    }
}

// one output of k_sum_parts (device function form: the same association)
template <int SL>
__device__ __forceinline__ void sum_parts_block(const float* __restrict__ part, int n_part, int n, int block, float (*sR)[256 / SL + 1], int& i,
                                                float& sum, bool& writer) {
    constexpr int NO = 256 / SL;
    const int o = threadIdx.x % NO, sl = threadIdx.x / NO;
    i = block * NO + o;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < n) {
        int k = sl;
        for (; k + 3 * SL < n_part; k += 4 * SL) {
            acc[0] += part[(size_t)k * n + i];
            acc[1] += part[(size_t)(k + SL) * n + i];
            acc[2] += part[(size_t)(k + 2 * SL) * n + i];
            acc[3] += part[(size_t)(k + 3 * SL) * n + i];
        }
        for (; k < n_part; k += SL) acc[0] += part[(size_t)k * n + i];
    }
    sR[sl][o] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    sum = 0.f;
    writer = sl == 0 && i < n;
    if (writer) {
#pragma unroll
        for (int q = 0; q < SL; ++q) sum += sR[q][o];
    }
}
// The two sums behind a weight-gradient launch whose thin pieces live in per-workgroup partials (k_sum_parts<4> over the chunk slabs,
// k_sum_parts<16> over the thin partials) in ONE launch (round 5: two dependent 5 - 10 us launches per network gradient at large
// batches).  Blocks [0, nb_slab): the hidden kernel's entries from the slabs (the slabs' thin positions are not read); the rest: the
// thin positions from the partials.  The same association per output as the two launches: bit-identical gradients.
__global__ void __launch_bounds__(256) k_sum_slabs_thin(const float* __restrict__ slabs, int n_slab, int n, int w2_off, int nb_slab,
                                                        const float* __restrict__ thin, int n_thin_part, int n_thin, float* __restrict__ out,
                                                        const FinishJob fin, int nb_all) {
    __shared__ float sR4[4][65];
    __shared__ float sR16[16][17];
    int i;
    float sum;
    bool writer;
    if ((int)blockIdx.x == nb_all) {          // the extra block: a pending scalar reduction (k_finish_parts' arithmetic, block order)
        if (threadIdx.x == 0) {
            float sacc = 0.f;
            for (int b = 0; b < fin.n_part; ++b) sacc += fin.part[b];
            fin.out0[0] = fin.scale0 * sacc;
        } else if (threadIdx.x == 64 && fin.out1) {
            float sacc = 0.f;
            for (int b = 0; b < fin.n_part; ++b) sacc += fin.part[fin.stride1 + b];
            fin.out1[0] = fin.scale1 * sacc;
        }
        return;
    }
    if ((int)blockIdx.x < nb_slab) {
        // (slab column = position in the flat network: the W2 block starts at w2_off)
        sum_parts_block<4>(slabs + w2_off, n_slab, n, blockIdx.x, sR4, i, sum, writer);
        if (writer && i < H * H) out[w2_off + i] = sum;
    } else {
        sum_parts_block<16>(thin, n_thin_part, n_thin, (int)blockIdx.x - nb_slab, sR16, i, sum, writer);
        if (writer) out[i < w2_off ? i : i + H * H] = sum;
    }
}

int launch_thin_reduce(const float* part, int n_part, int in_dim, int out_dim, float* grad, hipStream_t s) {
    const int n_thin = thin_floats(in_dim, out_dim);
    // thin positions of the flat gradient: W1 | b1 in front of W2, b2 | W3 | b3 behind it
    hipLaunchKernelGGL((k_sum_parts<16>), dim3((n_thin + 15) / 16), dim3(256), 0, s, part, n_part, n_thin, in_dim * H + H, H * H, grad);
    MPG_CHECK_LAUNCH("k_sum_parts (thin)");
    return MPG_OK;
}

template <int I, int O>
static void launch_backward_thin(const BwdArgs& a, int grid, hipStream_t s) {
    if constexpr (I <= 8) {
        if (a.pack) hipLaunchKernelGGL((k_backward<I, O, false, true, true>), dim3(grid), dim3(NTHREAD), 0, s, a);
        else hipLaunchKernelGGL((k_backward<I, O, false, false, true>), dim3(grid), dim3(NTHREAD), 0, s, a);
    }
}

bool backward_takes_thin(int in_dim, int ou) {
    return in_dim <= 8 && !(in_dim == 7 && ou == 2) && !(in_dim == 8 && ou == 2);
}
int backward_thin_parts(int rows) { return grid_for((rows + GROUP - 1) / GROUP); }

int launch_backward(const mpg_cfg_t* cfg, const float* params, int in_dim, int out_dim, int ou, int rows, const float* dy, int lddy,
                    const float* yout, int ldyo, int out_tanh, float out_scale, const float* h1, const float* h2,
                    float* dz1, float* dz2, float* dz3, float* dx, int lddx, hipStream_t s, const XSpec* thin_x, float* thin_part) {
    MPG_REQUIRE(params && rows > 0 && dy && h1 && h2 && (!out_tanh || yout), "launch_backward: bad argument");
    MPG_REQUIRE(!thin_part || (thin_x && !dx && backward_takes_thin(in_dim, ou) && thin_x->d0 + thin_x->d1 == in_dim),
                "launch_backward: thin gradients need the network input, no dx and a base input width");
    BwdArgs a;
    a.thin_part = thin_part;
    if (thin_x) a.x = *thin_x;
    a.params = params; a.in_dim = in_dim; a.out_dim = out_dim; a.rows = rows; a.dy = dy; a.lddy = lddy;
    a.yout = yout; a.ldyo = ldyo; a.out_tanh = out_tanh; a.out_scale = out_scale; a.h1 = h1; a.h2 = h2;
    a.dz1 = dz1; a.dz2 = dz2; a.dz3 = dz3; a.dx = dx; a.lddx = lddx;
    a.pack = weight_cache_lookup(cfg, make_net(params, in_dim, out_dim).W2, 1);
    const long ngroups = (rows + GROUP - 1) / GROUP;
    mpg_prof_begin(mpg_prof_of(cfg), 4, s);
    if (dx) {
#define CALL(I, O)                                                                                                \
    if (a.pack) hipLaunchKernelGGL((k_backward<I, O, true, true>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, a); \
    else hipLaunchKernelGGL((k_backward<I, O, true, false>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, a)
        MPG_DISPATCH_NET(in_dim, ou, CALL)
#undef CALL
    } else if (thin_part) {
#define CALL(I, O) launch_backward_thin<I, O>(a, grid_for(ngroups), s)
        MPG_DISPATCH_NET(in_dim, ou, CALL)
#undef CALL
    } else {
#define CALL(I, O)                                                                                                 \
    if (a.pack) hipLaunchKernelGGL((k_backward<I, O, false, true>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, a); \
    else hipLaunchKernelGGL((k_backward<I, O, false, false>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, a)
        MPG_DISPATCH_NET(in_dim, ou, CALL)
#undef CALL
    }
    mpg_prof_end(mpg_prof_of(cfg), 4, s);
    MPG_CHECK_LAUNCH("k_backward");
    return MPG_OK;
}

// -------------------------------------------------------------------------------------------------------
// weight gradient:  dW2 = H1^T DZ2 on MFMA straight from the G16 stashes (k = batch row), the thin pieces
// (dW1, db1, db2, dW3, db3) on VALU.  Grid = (8 column slices, chunks of row groups): a workgroup accumulates the
// 256 x 32 column slice of dW2 over its chunk (wave w: feature tiles 2w, 2w+1), so a chunk's slab is written once
// by 8 workgroups; k_sum_parts sums the chunk slabs in a fixed order (deterministic, no float atomics).
// -------------------------------------------------------------------------------------------------------
template <int IN, int OU>
__global__ void __launch_bounds__(NTHREAD, IN <= 8 ? 4 : 2) k_wgrad(const WgradArgs a) {      // (16-wide: 80 KB of LDS, one workgroup per CU anyway)
    __shared__ __attribute__((aligned(16))) float sRed[NWAVE * wgrad_nq<IN, OU>() * 64];
    int chunk, sl;
    wgrad_map(blockIdx.x, gridDim.x >> 3, chunk, sl);
    wgrad_body<IN, OU>(a, sl, chunk, sRed);
}

#ifdef MPG_SPLIT
// dW2 only (the thin pieces were accumulated elsewhere: launch_wgrad no_thin), 64-column slices: four workgroups per chunk re-read the
// chunk's H1 through L2 instead of eight
template <int IN, int OU>
__global__ void __launch_bounds__(NTHREAD, 4) k_wgrad_w2(const WgradArgs a) {
    __shared__ __attribute__((aligned(16))) float sRed[4 * 4 * 2 * 64 * 2 * 2];   // the B tile: 4 pairs x 4 tiles x (hi, lo) x 64 lanes x 2 groups x 8 bytes
    int chunk, sl;
    wgrad_map4(blockIdx.x, gridDim.x >> 2, chunk, sl);
    wgrad_body<IN, OU, 1, 4>(a, sl, chunk, sRed);
}
#endif

#ifdef MPG_SPLIT
template <int I, int O>
static void launch_wgrad_w2(const WgradArgs& a, int nch, hipStream_t s) {
    if constexpr (I <= 8) hipLaunchKernelGGL((k_wgrad_w2<I, O>), dim3(4 * nch), dim3(NTHREAD), 0, s, a);
}
#endif

size_t wgrad_workspace_floats(int rows, int in_dim, int out_dim) {
    const long ngroups = (rows + GROUP - 1) / GROUP;
    // the single-job launch cuts finer than the multi-job one: room for whichever is used
    const int gp = wgrad_groups_per_chunk(ngroups, true), gpm = wgrad_groups_per_chunk(ngroups, false), gpw = wgrad_groups_per_chunk_w2(ngroups);
    const long nch = std::max(std::max((ngroups + gp - 1) / gp, (ngroups + gpm - 1) / gpm), (ngroups + gpw - 1) / gpw);
    return (size_t)nch * net_size(in_dim, out_dim);
}

int launch_wgrad(const mpg_cfg_t* cfg, int in_dim, int out_dim, int ou, int rows, const XSpec& x, const float* h1, const float* h2,
                 const float* dz1, const float* dz2, const float* dz3, float inv_b, float* grad, float* ws, hipStream_t s, bool no_thin,
                 const float* thin_part, int n_thin_part, const FinishJob* fin) {
    MPG_REQUIRE(rows > 0 && h1 && h2 && (dz1 || no_thin) && dz2 && dz3 && grad && ws, "launch_wgrad: bad argument");
    WgradArgs a;
    a.no_thin = no_thin ? 1 : 0;
    a.in_dim = in_dim; a.out_dim = out_dim; a.rows = rows; a.x = x;
    a.h1 = h1; a.h2 = h2; a.dz1 = dz1; a.dz2 = dz2; a.dz3 = dz3; a.slabs = ws;
    (void)inv_b;
    const long ngroups = (rows + GROUP - 1) / GROUP;
#if defined(MPG_SPLIT)
    const bool w2_only = no_thin && backward_takes_thin(in_dim, ou);       // (the base input widths: where the thin pieces can live elsewhere)
#else
    const bool w2_only = false;
#endif
    a.groups_per_chunk = w2_only ? wgrad_groups_per_chunk_w2(ngroups) : wgrad_groups_per_chunk(ngroups, true);
    const int nch = (int)((ngroups + a.groups_per_chunk - 1) / a.groups_per_chunk);
    mpg_prof_begin(mpg_prof_of(cfg), 5, s);
    if (w2_only) {
#ifdef MPG_SPLIT
#define CALL(I, O) launch_wgrad_w2<I, O>(a, nch, s)
        MPG_DISPATCH_NET(in_dim, ou, CALL)
#undef CALL
#endif
    } else {
#define CALL(I, O) hipLaunchKernelGGL((k_wgrad<I, O>), dim3(8 * nch), dim3(NTHREAD), 0, s, a)
    MPG_DISPATCH_NET(in_dim, ou, CALL)
#undef CALL
    }
    mpg_prof_end(mpg_prof_of(cfg), 5, s);
    MPG_CHECK_LAUNCH("k_wgrad");
    const int n = net_size(in_dim, out_dim);
    if (no_thin && thin_part) {        // the chunk slabs (hidden kernel) and the thin partials of the backward launch / reverse sweep in one launch
        const int n_thin = thin_floats(in_dim, out_dim), nb_slab = H * H / 64, nb_all = nb_slab + (n_thin + 15) / 16;
        hipLaunchKernelGGL(k_sum_slabs_thin, dim3(nb_all + (fin ? 1 : 0)), dim3(256), 0, s, ws, nch, n, in_dim * H + H, nb_slab, thin_part,
                           n_thin_part, n_thin, grad, fin ? *fin : FinishJob{}, nb_all);
        MPG_CHECK_LAUNCH("k_sum_slabs_thin");
        return MPG_OK;
    }
    MPG_REQUIRE(!fin, "launch_wgrad: a pending scalar reduction needs the merged summation launch");
    hipLaunchKernelGGL((k_sum_parts<4>), dim3((n + 63) / 64), dim3(256), 0, s, ws, nch, n, n, 0, grad);     // the chunk slabs
    MPG_CHECK_LAUNCH("k_sum_parts (slabs)");
    return MPG_OK;
}

}  // namespace mlp
