// K3/K5: generic network kernels on the weight-stationary engine of mlp_core.h:
//   forward (optionally stashing hidden activations), input-side backward, weight gradient (+ slab reduce).
#include "mlp_launch.h"

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
};

// loads one row group of the network input into sX [16][XS] (zero padded)
template <int IN>
__device__ __forceinline__ void load_x_group(const XSpec& x, int rows, long g, float* sX) {
    const int tid = threadIdx.x;
    if (tid < GROUP * XS) {
        const int row = tid / XS, i = tid % XS;
        const long gr = g * GROUP + row;
        float v = 0.f;
        if (gr < rows && i < IN) {
            if (i < x.d0)
                v = x.x0[gr * x.ld0 + i] * x.scale[i];
            else
                v = x.x1[gr * x.ld1 + (i - x.d0)];
        }
        sX[tid] = v;
    }
}

template <int IN, int OU>
__global__ void __launch_bounds__(NTHREAD, 2) k_forward(const FwdArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[GROUP * LDA + GROUP * XS + NWAVE * GROUP * MAXOUT];
    float* sA = smem;
    float* sX = sA + GROUP * LDA;
    float* sPart = sX + GROUP * XS;
    const Lane L;
    const Net net = make_net(a.params, a.in_dim, a.out_dim);
    float w2[128];
    SmallRegs<IN, OU> r;
    if (a.pack) load_w2_packed(a.pack, L, w2); else load_w2_fwd(net.W2, L, w2);
    load_small<IN, OU>(net, L, r);
    const long ngroups = (a.rows + GROUP - 1) / GROUP;
    for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        load_x_group<IN>(a.x, a.rows, g, sX);
        lds_barrier();
        float h1[2][4], h2[2][4];
        forward_group<IN, OU>(sX, sA, sPart, L, w2, r, h1, h2);
        if (a.h1) stash_store(a.h1, g, L, h1);
        if (a.h2) stash_store(a.h2, g, L, h2);
        const int tid = threadIdx.x;
        if (tid < GROUP * OU) {
            const int row = tid / OU, o = tid % OU;
            const long gr = g * GROUP + row;
            if (gr < a.rows) {
                float z = out_preact(sPart, net.b3[o], row, o);
                float y = a.out_tanh ? a.out_scale * tanhf(z) : z;
                if (a.sigma > 0.f) {   // OffPolicyWorker.sample: action += N(0, sigma), worker.py:97-98
                    Philox4 p = philox4x32_10((uint32_t)gr, a.c1, a.c2, 0x5eedu + (uint32_t)o, a.k0, a.k1);
                    float u1 = u01(p.v[0]), u2 = u01(p.v[1]);
                    y += a.sigma * sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
                }
                a.y[gr * a.ldy + o] = y;
            }
        }
    }
}

#define MPG_DISPATCH_NET(in_dim, ou, CALL)                                    \
    if ((in_dim) == 6 && (ou) == 2) { CALL(6, 2); }                           \
    else if ((in_dim) == 8 && (ou) == 1) { CALL(8, 1); }                      \
    else if ((in_dim) == 4 && (ou) == 1) { CALL(4, 1); }                      \
    else if ((in_dim) == 5 && (ou) == 1) { CALL(5, 1); }                      \
    else if ((in_dim) == 6 && (ou) == 1) { CALL(6, 1); }                      \
    else {                                                                    \
        mpg_set_error("unsupported network shape in=%d used-out=%d", (in_dim), (ou)); \
        return MPG_EINVAL;                                                    \
    }

static int grid_for(long ngroups) { return (int)(ngroups < 256 ? ngroups : 256); }

int launch_forward(const float* params, int in_dim, int out_dim, int ou, int rows, const XSpec& x, const OutSpec& o,
                   float* y, int ldy, float* h1, float* h2, hipStream_t s) {
    MPG_REQUIRE(params && rows > 0 && y && x.d0 + x.d1 == in_dim && ou <= out_dim, "launch_forward: bad argument");
    FwdArgs a;
    a.params = params; a.in_dim = in_dim; a.out_dim = out_dim; a.rows = rows; a.x = x;
    a.out_tanh = o.out_tanh; a.out_scale = o.out_scale; a.sigma = o.sigma;
    a.k0 = (uint32_t)o.seed; a.k1 = (uint32_t)(o.seed >> 32); a.c1 = (uint32_t)o.ctr; a.c2 = (uint32_t)(o.ctr >> 32);
    a.y = y; a.ldy = ldy; a.h1 = h1; a.h2 = h2;
    a.pack = weight_cache_lookup(make_net(params, in_dim, out_dim).W2, 0);
    const long ngroups = (rows + GROUP - 1) / GROUP;
    mpg_prof_begin(3, s);
#define CALL(I, O) hipLaunchKernelGGL((k_forward<I, O>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, a)
    MPG_DISPATCH_NET(in_dim, ou, CALL)
#undef CALL
    mpg_prof_end(3, s);
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
};

template <int IN, int OU, bool WANT_DX>
__global__ void __launch_bounds__(NTHREAD, 2) k_backward(const BwdArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[GROUP * LDA + GROUP * MAXOUT + NWAVE * GROUP * XS];
    float* sA = smem;
    float* sD3 = sA + GROUP * LDA;
    float* sPartX = sD3 + GROUP * MAXOUT;
    const Lane L;
    const Net net = make_net(a.params, a.in_dim, a.out_dim);
    float w2t[128];
    SmallRegs<IN, OU> r;
    if (a.pack) load_w2_packed(a.pack, L, w2t); else load_w2_bwd(net.W2, L, w2t);
    load_small<IN, OU>(net, L, r);
    const long ngroups = (a.rows + GROUP - 1) / GROUP;
    const int tid = threadIdx.x;
    for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        if (tid < GROUP * OU) {
            const int row = tid / OU, o = tid % OU;
            const long gr = g * GROUP + row;
            float d = 0.f;
            if (gr < a.rows) {
                d = a.dy[gr * a.lddy + o];
                if (a.out_tanh) {   // a = S*tanh(z): da/dz = S*(1 - (a/S)^2)
                    const float t = a.yout[gr * a.ldyo + o] / a.out_scale;
                    d *= a.out_scale * (1.f - t * t);
                }
                if (a.dz3) a.dz3[gr * OU + o] = d;
            }
            sD3[row * MAXOUT + o] = d;
        }
        float h1[2][4], h2[2][4], dz1[2][4], dz2[2][4];
        stash_load(a.h1, g, L, h1);
        stash_load(a.h2, g, L, h2);
        lds_barrier();
        backward_group<IN, OU, WANT_DX>(sD3, sA, sPartX, L, w2t, r, h1, h2, dz1, dz2);
        if (a.dz1) stash_store(a.dz1, g, L, dz1);
        if (a.dz2) stash_store(a.dz2, g, L, dz2);
        if (WANT_DX) {
            if (tid < GROUP * IN) {
                const int row = tid / IN, i = tid % IN;
                const long gr = g * GROUP + row;
                if (gr < a.rows) a.dx[gr * a.lddx + i] = dx_reduce(sPartX, row, i);
            }
            lds_barrier();   // sPartX / sD3 are rewritten by the next group
        }
    }
}

int launch_backward(const float* params, int in_dim, int out_dim, int ou, int rows, const float* dy, int lddy,
                    const float* yout, int ldyo, int out_tanh, float out_scale, const float* h1, const float* h2,
                    float* dz1, float* dz2, float* dz3, float* dx, int lddx, hipStream_t s) {
    MPG_REQUIRE(params && rows > 0 && dy && h1 && h2 && (!out_tanh || yout), "launch_backward: bad argument");
    BwdArgs a;
    a.params = params; a.in_dim = in_dim; a.out_dim = out_dim; a.rows = rows; a.dy = dy; a.lddy = lddy;
    a.yout = yout; a.ldyo = ldyo; a.out_tanh = out_tanh; a.out_scale = out_scale; a.h1 = h1; a.h2 = h2;
    a.dz1 = dz1; a.dz2 = dz2; a.dz3 = dz3; a.dx = dx; a.lddx = lddx;
    a.pack = weight_cache_lookup(make_net(params, in_dim, out_dim).W2, 1);
    const long ngroups = (rows + GROUP - 1) / GROUP;
    mpg_prof_begin(4, s);
    if (dx) {
#define CALL(I, O) hipLaunchKernelGGL((k_backward<I, O, true>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, a)
        MPG_DISPATCH_NET(in_dim, ou, CALL)
#undef CALL
    } else {
#define CALL(I, O) hipLaunchKernelGGL((k_backward<I, O, false>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, a)
        MPG_DISPATCH_NET(in_dim, ou, CALL)
#undef CALL
    }
    mpg_prof_end(4, s);
    MPG_CHECK_LAUNCH("k_backward");
    return MPG_OK;
}

// -------------------------------------------------------------------------------------------------------
// weight gradient:  dW2 = H1^T DZ2 on MFMA straight from the G16 stashes (k = batch row), the thin pieces
// (dW1, db1, db2, dW3, db3) on VALU.  Grid = (8 column slices, chunks of row groups): a workgroup accumulates the
// 256 x 32 column slice of dW2 over its chunk (wave w: feature tiles 2w, 2w+1), so a chunk's slab is written once
// by 8 workgroups; k_reduce_slabs sums the <= 32 chunk slabs in a fixed order (deterministic, no float atomics).
// -------------------------------------------------------------------------------------------------------
struct WgradArgs {
    int in_dim, out_dim, rows, groups_per_chunk;
    XSpec x;
    const float *h1, *h2, *dz1, *dz2, *dz3;
    float* slabs;
};

template <int IN>
__device__ __forceinline__ float x_value(const XSpec& x, long gr, int i) {
    return i < x.d0 ? x.x0[gr * x.ld0 + i] * x.scale[i] : x.x1[gr * x.ld1 + (i - x.d0)];
}

template <int IN, int OU>
__global__ void __launch_bounds__(NTHREAD, 2) k_wgrad(const WgradArgs a) {
    constexpr int NQ = 2 * IN + 4 + 2 * OU + OU;      // thin quantities per lane
    __shared__ float sRed[NWAVE * NQ * 64];
    const Lane L;
    const int tid = threadIdx.x;
    const int sl = blockIdx.x;                         // column slice: hidden columns [32 sl, 32 sl + 32)
    const long ngroups = (a.rows + GROUP - 1) / GROUP;
    const long g0 = (long)blockIdx.y * a.groups_per_chunk;
    const long g1 = (g0 + a.groups_per_chunk < ngroups) ? g0 + a.groups_per_chunk : ngroups;
    f32x4 acc[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) acc[u][0] = acc[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    float gW1[2][IN], gb1[2] = {0.f, 0.f}, gb2[2] = {0.f, 0.f}, gW3[2][OU], gb3[OU];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int i = 0; i < IN; ++i) gW1[t][i] = 0.f;
#pragma unroll
        for (int o = 0; o < OU; ++o) gW3[t][o] = 0.f;
    }
#pragma unroll
    for (int o = 0; o < OU; ++o) gb3[o] = 0.f;
    const f32x4* H1 = reinterpret_cast<const f32x4*>(a.h1);
    const f32x4* H2 = reinterpret_cast<const f32x4*>(a.h2);
    const f32x4* DZ1 = reinterpret_cast<const f32x4*>(a.dz1);
    const f32x4* DZ2 = reinterpret_cast<const f32x4*>(a.dz2);
    for (long g = g0; g < g1; ++g) {
        const f32x4 b0 = DZ2[(g * 16 + 2 * sl) * 64 + L.lane], b1 = DZ2[(g * 16 + 2 * sl + 1) * 64 + L.lane];
        const f32x4 a0 = H1[(g * 16 + 2 * L.wave) * 64 + L.lane], a1 = H1[(g * 16 + 2 * L.wave + 1) * 64 + L.lane];
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // the float4's 4 entries are 4 k-steps (k = batch row)
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
        }
        if ((int)((g - g0) & 7) == L.wave) {   // thin pieces: the chunk's groups are dealt round-robin to the 8 waves
            const f32x4 d10 = DZ1[(g * 16 + 2 * sl) * 64 + L.lane], d11 = DZ1[(g * 16 + 2 * sl + 1) * 64 + L.lane];
            const f32x4 h20 = H2[(g * 16 + 2 * sl) * 64 + L.lane], h21 = H2[(g * 16 + 2 * sl + 1) * 64 + L.lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long gr = g * GROUP + L.row(j);
                const bool live = gr < a.rows;
                float d3[OU];
#pragma unroll
                for (int o = 0; o < OU; ++o) d3[o] = live ? a.dz3[gr * OU + o] : 0.f;
                gb1[0] += d10[j]; gb1[1] += d11[j];
                gb2[0] += b0[j];  gb2[1] += b1[j];
#pragma unroll
                for (int i = 0; i < IN; ++i) {
                    const float xv = live ? x_value<IN>(a.x, gr, i) : 0.f;
                    gW1[0][i] = fmaf(xv, d10[j], gW1[0][i]);
                    gW1[1][i] = fmaf(xv, d11[j], gW1[1][i]);
                }
#pragma unroll
                for (int o = 0; o < OU; ++o) {
                    gW3[0][o] = fmaf(h20[j], d3[o], gW3[0][o]);
                    gW3[1][o] = fmaf(h21[j], d3[o], gW3[1][o]);
                    if (L.c == 0) gb3[o] += d3[o];
                }
            }
        }
    }
    // ---- this workgroup's part of the chunk slab ----
    float* slab = a.slabs + (size_t)blockIdx.y * net_size(a.in_dim, a.out_dim);
    float* sW1 = slab;
    float* sb1 = sW1 + a.in_dim * H;
    float* sW2 = sb1 + H;
    float* sb2 = sW2 + H * H;
    float* sW3 = sb2 + H;
    float* sb3 = sW3 + H * a.out_dim;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                sW2[(16 * (2 * L.wave + u) + 4 * L.rg + j) * H + 32 * sl + 16 * t + L.c] = acc[u][t][j];
    // thin pieces: sum over the 8 waves and the 4 row quads through LDS in a fixed order
    {
        float* dst = sRed + (L.wave * NQ) * 64 + L.lane;
        int q = 0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int i = 0; i < IN; ++i) dst[(q++) * 64] = gW1[t][i];
            dst[(q++) * 64] = gb1[t];
            dst[(q++) * 64] = gb2[t];
#pragma unroll
            for (int o = 0; o < OU; ++o) dst[(q++) * 64] = gW3[t][o];
        }
#pragma unroll
        for (int o = 0; o < OU; ++o) dst[(q++) * 64] = gb3[o];
    }
    __syncthreads();
    for (int item = tid; item < NQ * 16; item += NTHREAD) {
        const int q = item / 16, c = item % 16;
        float sum = 0.f;
        for (int w = 0; w < NWAVE; ++w)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) sum += sRed[(w * NQ + q) * 64 + rg * 16 + c];
        constexpr int PER_T = IN + 2 + OU;
        if (q < 2 * PER_T) {
            const int t = q / PER_T, r = q % PER_T, col = 32 * sl + 16 * t + c;
            if (r < IN) sW1[r * H + col] = sum;
            else if (r == IN) sb1[col] = sum;
            else if (r == IN + 1) sb2[col] = sum;
            else sW3[col * a.out_dim + (r - IN - 2)] = sum;
        } else if (sl == 0) {
            // db3[o]: lanes with c == 0 carried it; summing over c adds exact zeros
            float tot = sum;
            tot += __shfl_xor(tot, 1, 16); tot += __shfl_xor(tot, 2, 16); tot += __shfl_xor(tot, 4, 16); tot += __shfl_xor(tot, 8, 16);
            if (c == 0) sb3[q - 2 * PER_T] = tot;
        }
    }
    // unused output columns of W3 / b3 (the log-std half of the policy head, SURVEY B-5) have zero gradient
    for (int item = tid; item < 32 * (a.out_dim - OU); item += NTHREAD) {
        const int col = 32 * sl + item / (a.out_dim - OU), o = OU + item % (a.out_dim - OU);
        sW3[col * a.out_dim + o] = 0.f;
    }
    if (sl == 0 && tid < a.out_dim - OU) sb3[OU + tid] = 0.f;
}

__global__ void k_reduce_slabs(const float* __restrict__ slabs, int nslab, int n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < nslab; ++k) s += slabs[(size_t)k * n + i];
    out[i] = s;
}

static int wgrad_groups_per_chunk(long ngroups) {
    long gp = (ngroups + 31) / 32;   // <= 32 chunk slabs
    return (int)(gp < 1 ? 1 : gp);
}

size_t wgrad_workspace_floats(int rows, int in_dim, int out_dim) {
    const long ngroups = (rows + GROUP - 1) / GROUP;
    const int gp = wgrad_groups_per_chunk(ngroups);
    const long nch = (ngroups + gp - 1) / gp;
    return (size_t)nch * net_size(in_dim, out_dim);
}

int launch_wgrad(int in_dim, int out_dim, int ou, int rows, const XSpec& x, const float* h1, const float* h2,
                 const float* dz1, const float* dz2, const float* dz3, float* grad, float* ws, hipStream_t s) {
    MPG_REQUIRE(rows > 0 && h1 && h2 && dz1 && dz2 && dz3 && grad && ws, "launch_wgrad: bad argument");
    WgradArgs a;
    a.in_dim = in_dim; a.out_dim = out_dim; a.rows = rows; a.x = x;
    a.h1 = h1; a.h2 = h2; a.dz1 = dz1; a.dz2 = dz2; a.dz3 = dz3; a.slabs = ws;
    const long ngroups = (rows + GROUP - 1) / GROUP;
    a.groups_per_chunk = wgrad_groups_per_chunk(ngroups);
    const int nch = (int)((ngroups + a.groups_per_chunk - 1) / a.groups_per_chunk);
    mpg_prof_begin(5, s);
#define CALL(I, O) hipLaunchKernelGGL((k_wgrad<I, O>), dim3(8, nch), dim3(NTHREAD), 0, s, a)
    MPG_DISPATCH_NET(in_dim, ou, CALL)
#undef CALL
    mpg_prof_end(5, s);
    MPG_CHECK_LAUNCH("k_wgrad");
    const int n = net_size(in_dim, out_dim);
    hipLaunchKernelGGL(k_reduce_slabs, dim3((n + 255) / 256), dim3(256), 0, s, ws, nch, n, grad);
    MPG_CHECK_LAUNCH("k_reduce_slabs");
    return MPG_OK;
}

}  // namespace mlp
