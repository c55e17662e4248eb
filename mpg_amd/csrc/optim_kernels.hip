// K7/K8: tf.clip_by_global_norm per network, Keras Adam and Polyak target updates - one launch each over the
// flat [net0 | net1 | ...] parameter vector.  Deterministic (fixed-order block reductions, no float atomics).
//
// Reference: learners/mpg_learner.py:415-431 (clip), policy.py:123-171 (apply_gradients, update_*_target),
// optimizer.py:357-361 (NaN guard).  Adam follows TensorFlow's ApplyAdam functor:
//   m += (g - m)(1 - b1);  v += (g^2 - v)(1 - b2);  w -= lr_t * m / (sqrt(v) + eps),  lr_t = lr sqrt(1-b2^t)/(1-b1^t).
#include "mlp_core.h"   // net_size, wcache_w2_offset


namespace {

constexpr int MAXSEG = 8;

struct Segs {
    int n_seg;
    int off[MAXSEG], n[MAXSEG];
    float lr_t[MAXSEG];
    int do_adam[MAXSEG], do_polyak[MAXSEG];
    int w2_off[MAXSEG];              // offset of W2 inside the segment's network, -1 when no weight cache is bound
    float *cache_w, *cache_t;        // packed register images of the bound buffers (nullable)
    int *status_w, *status_t;        // their owners' MPG_STATUS_* words (nullable)
};

constexpr int HH = MPG_HIDDEN * MPG_HIDDEN;
#ifdef MPG_SPLIT
// Packed images of the split engine (mlp_core.h / weight_cache.hip): the fp16 hi and lo halves of W2[row][col] * W_SCALE,
// where contraction index k owns the lane group and register, output index n the lane column.  Writes both halves.
__device__ __forceinline__ void pack_store(float* __restrict__ image, int k, int n, float w) {
    const int wave = n >> 5, t = (n >> 4) & 1, c = n & 15, kb = k >> 5, rg = (k >> 3) & 3, r = (k >> 1) & 3, e = k & 1;
    const int v = (kb * 2 + t) * 2;                                                       // hi block; the lo block is v + 1
    const int word = ((mlp::img_slot(wave, v) * 64 + rg * 16 + c) << 2) + r, word_lo = ((mlp::img_slot(wave, v + 1) * 64 + rg * 16 + c) << 2) + r;
    // a parameter beyond the engine's envelope (|w| >= 1023.5, include/mpg_hip.h) enters the image clamped, never as an fp16
    // infinity; the caller of pack_store reports it (range_check)
    const float ws = fminf(fmaxf(w * mlp::W_SCALE, -65504.f), 65504.f);
    const _Float16 hi = (_Float16)ws;
    const _Float16 lo = (_Float16)(ws - (float)hi);
    _Float16* p = reinterpret_cast<_Float16*>(image);
    p[2 * word + e] = hi;
    p[2 * word_lo + e] = lo;
}
#else
// position of W2[row][col] in the packed image where `col`-like index n owns the lane and `row`-like index k the step
__device__ __forceinline__ int pack_index(int k, int n) {
    const int wave = n >> 5, t = (n >> 4) & 1, c = n & 15, q = k >> 2, rg = k & 3;
    return ((((wave * 16 + (q >> 2)) * 2 + t) * 64 + rg * 16 + c) << 2) + (q & 3);
}
__device__ __forceinline__ void pack_store(float* __restrict__ image, int k, int n, float w) { image[pack_index(k, n)] = w; }
#endif

// every hidden-kernel (W2) and output-kernel (W3) entry the update writes is checked against the envelope of the split engine
// (one compare in a memory-bound kernel).  `e` = index relative to the network's W2 (-1: no weight cache bound): W2 is [0, HH),
// b2 [HH, HH + 256), W3 and b3 follow.  The first layer and the hidden biases have no fp16 envelope (they are float32 operands of
// the fp32 MFMA / the accumulator) and are not flagged; a NaN anywhere still is (the comparison is false).  The out_dim entries
// of b3 behind W3 are held to W3's limit as well (this check does not know out_dim; an output bias beyond 1023.5 is reported
// although the engine could carry it - stricter than necessary, never laxer).  e = -1 (no usable weight cache bound): only NaN
// is looked for - the callers pass a null status word in exactly that case (adam_args: status_w / status_t come from the
// cache descriptors), so nothing could be reported anyway; the envelope is then checked when a cache is next packed.
__device__ __forceinline__ void range_check(int* status, float w, int e) {
    const bool enveloped = (e >= 0 && e < HH) || e >= HH + MPG_HIDDEN;
    if (status && (enveloped ? !(fabsf(w) < mlp::P_LIMIT) : w != w)) atomicOr(status, MPG_STATUS_PARAMETER_RANGE);
}

// Parallel form of the clip.  (1) per-network partial sums of squares, one per 256-element block, MPG_CLIP_PARTS slots
// per network (`k_sq_blocks`; the fused gradient kernel's final slab reduction writes the SAME partials for free, see
// mpg_block_sum256 in mpg_common.h); (2) every consumer rebuilds its network's norm from the partials in one fixed order
// (seg_sumsq) - so the norm is bit-identical whichever kernel produced the partials.
constexpr int CLIP_PARTS = MPG_CLIP_PARTS;
__global__ void __launch_bounds__(256) k_sq_blocks(const Segs sg, const float* __restrict__ grad, float* __restrict__ part) {
    __shared__ float red[256];
    const int k = blockIdx.y, b = blockIdx.x;
    const float* g = grad + sg.off[k];
    const int n = sg.n[k];
    float a = 0.f;
    for (int i = b * 256 + threadIdx.x; i < n; i += CLIP_PARTS * 256) a = fmaf(g[i], g[i], a);   // one term when n <= 69 632
    const float tot = mpg_block_sum256(a, red);
    if (threadIdx.x == 0) part[k * CLIP_PARTS + b] = tot;
}

// sum of a network's CLIP_PARTS partials, executed by ONE wave (all 64 lanes return the total); fixed association
__device__ __forceinline__ float seg_sumsq(const float* __restrict__ part, int k) {
    const int lane = threadIdx.x & 63;
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < (CLIP_PARTS + 63) / 64; ++j) {
        const int i = lane + 64 * j;
        a += i < CLIP_PARTS ? part[k * CLIP_PARTS + i] : 0.f;
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) a += __shfl_xor(a, m, 64);
    return a;
}

__global__ void __launch_bounds__(256) k_clip_scale(const Segs sg, float* __restrict__ grad, const float* __restrict__ part,
                                                    float clip, float* __restrict__ norms, int* __restrict__ nonfinite) {
    __shared__ float s_scale;
    const int k = blockIdx.y;
    if (threadIdx.x < 64) {
        const float nrm = sqrtf(seg_sumsq(part, k));
        if (threadIdx.x == 0) {
            s_scale = clip * fminf(1.f / nrm, 1.f / clip);
            if (blockIdx.x == 0) {
                norms[k] = nrm;
                if (nonfinite) nonfinite[k] = isfinite(nrm) ? 0 : 1;
            }
        }
    }
    __syncthreads();
    const float sc = s_scale;
    float* g = grad + sg.off[k];
    const int i0 = blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 256;
        if (i < sg.n[k]) g[i] *= sc;
    }
}

// one block per segment: norm = sqrt(sum g^2); g *= clip * min(1/norm, 1/clip)   (tf.clip_by_global_norm)
// 8 independent accumulators / float4 loads keep ~8 loads in flight per lane (a serial fma chain on one load per
// iteration is L2-latency bound: 35 us for 68 k floats).
__global__ void __launch_bounds__(1024) k_clip(const Segs sg, float* __restrict__ grad, float clip,
                                               float* __restrict__ norms, int* __restrict__ nonfinite) {
    __shared__ float red[1024];
    __shared__ float s_scale;
    const int k = blockIdx.x;
    float* g = grad + sg.off[k];
    const int n = sg.n[k];
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int i = threadIdx.x;
    for (; i + 7 * 1024 < n; i += 8 * 1024) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = g[i + u * 1024];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = fmaf(v[u], v[u], acc[u]);
    }
    for (; i < n; i += 1024) acc[0] = fmaf(g[i], g[i], acc[0]);
    red[threadIdx.x] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float nrm = sqrtf(red[0]);
        norms[k] = nrm;
        s_scale = clip * fminf(1.f / nrm, 1.f / clip);
        if (nonfinite) nonfinite[k] = isfinite(nrm) ? 0 : 1;
    }
    __syncthreads();
    const float sc = s_scale;
    i = threadIdx.x;
    for (; i + 7 * 1024 < n; i += 8 * 1024) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = g[i + u * 1024];
#pragma unroll
        for (int u = 0; u < 8; ++u) g[i + u * 1024] = v[u] * sc;
    }
    for (; i < n; i += 1024) g[i] *= sc;
}

// Adam (Keras form, bias correction folded into lr_t; policy.py:123-156) and the Polyak mix (policy.py:158-171) for one
// element.  Contraction is off inside: k_adam_polyak and k_clip_adam_polyak must round identically whatever else surrounds
// the call (the native step driver is tested against the method-by-method path to 1e-6 over many iterations), and one
// rounding per written operation is also what the reference's float32 tensors do.
__device__ __forceinline__ void adam_update(float g, float lr_t, float& mj, float& vj, float& wj) {
#pragma clang fp contract(off)
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-7f;
    mj += (g - mj) * (1.f - b1);
    vj += (g * g - vj) * (1.f - b2);
    wj -= lr_t * mj / (sqrtf(vj) + eps);
}
__device__ __forceinline__ float polyak_mix(float tau, float wj, float tj) {
#pragma clang fp contract(off)
    return tau * wj + (1.f - tau) * tj;
}

__global__ void k_adam_polyak(const Segs sg, float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                              float* __restrict__ target, const float* __restrict__ grad, float tau,
                              const int* __restrict__ skip, int n_skip) {
    const int k = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sg.n[k]) return;
    const int j = sg.off[k] + i;
    float wj = w[j];
    const int e = sg.w2_off[k] >= 0 ? i - sg.w2_off[k] : -1;  // index relative to the network's W2 (weight cache, range_check)
    if (sg.do_adam[k]) {
        bool bad = false;                                     // optimizer.py:357-361: if ANY gradient is non-finite,
        if (skip)                                             // the whole list is replaced by zeros
            for (int q = 0; q < n_skip; ++q) bad |= skip[q] != 0;
        float mj = m[j], vj = v[j];
        adam_update(bad ? 0.f : grad[j], sg.lr_t[k], mj, vj, wj);
        m[j] = mj; v[j] = vj; w[j] = wj;
        range_check(sg.status_w, wj, e);
    }
    // weight cache: the element's two packed copies are rewritten by the thread that owns it
    const bool in_w2 = e >= 0 && e < HH;
    if (in_w2 && sg.do_adam[k] && sg.cache_w) {
        const int row = e >> 8, col = e & 255;
        pack_store(sg.cache_w + (size_t)(2 * k) * HH, row, col, wj);           // forward image: k = row, n = col
        pack_store(sg.cache_w + (size_t)(2 * k + 1) * HH, col, row, wj);       // backward image: k = col, n = row
    }
    if (sg.do_polyak[k] && target) {
        const float tj = polyak_mix(tau, wj, target[j]);
        target[j] = tj;
        range_check(sg.status_t, tj, e);
        if (in_w2 && sg.cache_t) {
            const int row = e >> 8, col = e & 255;
            pack_store(sg.cache_t + (size_t)(2 * k) * HH, row, col, tj);
            pack_store(sg.cache_t + (size_t)(2 * k + 1) * HH, col, row, tj);
        }
    }
}

// second half of the clip + Adam + Polyak for one element
__device__ __forceinline__ void clip_adam_polyak_element(const Segs& sg, int k, int i, int j, bool adam, bool polyak, float g_in, float wj,
                                                         float mj, float vj, float t_in, float nrm, bool bad, float clip, float tau,
                                                         float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                         float* __restrict__ target, float* __restrict__ grad) {
    const float gc = g_in * (clip * fminf(1.f / nrm, 1.f / clip));
    grad[j] = gc;
    const int e = sg.w2_off[k] >= 0 ? i - sg.w2_off[k] : -1;
    if (adam) {
        adam_update(bad ? 0.f : gc, sg.lr_t[k], mj, vj, wj);
        m[j] = mj; v[j] = vj; w[j] = wj;
        range_check(sg.status_w, wj, e);
    }
    const bool in_w2 = e >= 0 && e < HH;
    if (in_w2 && sg.do_adam[k] && sg.cache_w) {
        const int row = e >> 8, col = e & 255;
        pack_store(sg.cache_w + (size_t)(2 * k) * HH, row, col, wj);
        pack_store(sg.cache_w + (size_t)(2 * k + 1) * HH, col, row, wj);
    }
    if (polyak) {
        const float tj = polyak_mix(tau, wj, t_in);
        target[j] = tj;
        range_check(sg.status_t, tj, e);
        if (in_w2 && sg.cache_t) {
            const int row = e >> 8, col = e & 255;
            pack_store(sg.cache_t + (size_t)(2 * k) * HH, row, col, tj);
            pack_store(sg.cache_t + (size_t)(2 * k + 1) * HH, col, row, tj);
        }
    }
}

// clip (from the partials) + Adam + Polyak in one launch: mpg_clip_adam_polyak.  Same arithmetic as
// k_clip_scale followed by k_adam_polyak (the clipped gradient is also written back: it is the list the learner returns).
__global__ void __launch_bounds__(256) k_clip_adam_polyak(const Segs sg, float* __restrict__ w, float* __restrict__ m,
                                                          float* __restrict__ v, float* __restrict__ target,
                                                          float* __restrict__ grad, const float* __restrict__ part, float clip,
                                                          float tau, float* __restrict__ norms, int* __restrict__ nonfinite) {
    __shared__ float s_norm[MAXSEG];
    const int k = blockIdx.y;
    const int wave = threadIdx.x >> 6;
    // this thread's element first: its five loads travel together with the partials' (one memory round trip, not two)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < sg.n[k];
    const int j = sg.off[k] + (live ? i : 0);
    const bool adam = sg.do_adam[k] != 0, polyak = sg.do_polyak[k] && target;
    const float g_in = live ? grad[j] : 0.f;
    float wj = live ? w[j] : 0.f;
    float mj = (live && adam) ? m[j] : 0.f, vj = (live && adam) ? v[j] : 0.f;
    const float t_in = (live && polyak) ? target[j] : 0.f;
    for (int q = wave; q < sg.n_seg; q += 4) {
        const float nrm = sqrtf(seg_sumsq(part, q));
        if ((threadIdx.x & 63) == 0) s_norm[q] = nrm;
    }
    __syncthreads();
    bool bad = false;                                         // optimizer.py:357-361
    for (int q = 0; q < sg.n_seg; ++q) bad |= !isfinite(s_norm[q]);
    const float nrm = s_norm[k];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        norms[k] = nrm;
        if (nonfinite) nonfinite[k] = isfinite(nrm) ? 0 : 1;
    }
    if (!live) return;
    clip_adam_polyak_element(sg, k, i, j, adam, polyak, g_in, wj, mj, vj, t_in, nrm, bad, clip, tau, w, m, v, target, grad);
}

int fill(Segs& sg, int n_seg, const int* seg_sizes) {
    if (n_seg <= 0 || n_seg > MAXSEG || !seg_sizes) return -1;
    sg.n_seg = n_seg;
    int off = 0;
    for (int k = 0; k < MAXSEG; ++k) {
        sg.off[k] = off;
        sg.n[k] = k < n_seg ? seg_sizes[k] : 0;
        off += sg.n[k];
        sg.lr_t[k] = 0.f;
        sg.do_adam[k] = sg.do_polyak[k] = 0;
        sg.w2_off[k] = -1;
    }
    sg.cache_w = sg.cache_t = nullptr;
    sg.status_w = sg.status_t = nullptr;
    return off;
}

}  // namespace

extern "C" int mpg_clip_by_global_norm(float* grad, const int* seg_sizes, int n_seg, float clip, float* norms,
                                       int* nonfinite_flag, float* scratch, mpg_stream_t stream) {
    Segs sg;
    MPG_REQUIRE(grad && norms && fill(sg, n_seg, seg_sizes) > 0 && clip > 0.f, "mpg_clip_by_global_norm: bad argument");
    if (!scratch) {   // single-block-per-network form
        hipLaunchKernelGGL(k_clip, dim3(n_seg), dim3(1024), 0, mpg_stream(stream), sg, grad, clip, norms, nonfinite_flag);
        MPG_CHECK_LAUNCH("k_clip");
        return MPG_OK;
    }
    int maxn = 0;
    for (int k = 0; k < n_seg; ++k) maxn = sg.n[k] > maxn ? sg.n[k] : maxn;
    hipLaunchKernelGGL(k_sq_blocks, dim3(CLIP_PARTS, n_seg), dim3(256), 0, mpg_stream(stream), sg, grad, scratch);
    MPG_CHECK_LAUNCH("k_sq_blocks");
    hipLaunchKernelGGL(k_clip_scale, dim3((maxn + 1023) / 1024, n_seg), dim3(256), 0, mpg_stream(stream), sg, grad, scratch, clip,
                       norms, nonfinite_flag);
    MPG_CHECK_LAUNCH("k_clip_scale");
    return MPG_OK;
}

namespace {
// per-network Adam/Polyak switches + the packed register images of bound buffers (kept in sync inside the same kernel)
int fill_adam(Segs& sg, int n_seg, const int* seg_sizes, const float* w, const float* target, const float* lr_t,
              const int* do_adam, const int* do_polyak, const mpg_wcache_t* wc_w, const mpg_wcache_t* wc_t, int* maxn_out) {
    if (fill(sg, n_seg, seg_sizes) <= 0) return -1;
    int maxn = 0;
    for (int k = 0; k < n_seg; ++k) {
        sg.lr_t[k] = lr_t[k];
        sg.do_adam[k] = do_adam[k];
        sg.do_polyak[k] = do_polyak[k];
        if (sg.n[k] > maxn) maxn = sg.n[k];
    }
    *maxn_out = maxn;
    // a descriptor is honoured only if it describes exactly this buffer and this segmentation
    auto usable = [&](const mpg_wcache_t* wc, const float* base) {
        if (!wc || !base || wc->params != base || !wc->packed || wc->n_nets != n_seg) return false;
        for (int k = 0; k < n_seg; ++k)
            if (mlp::net_size(wc->in_dim[k], wc->out_dim[k]) != sg.n[k]) return false;
        return true;
    };
    const mpg_wcache_t* ref = usable(wc_w, w) ? wc_w : (usable(wc_t, target) ? wc_t : nullptr);
    if (usable(wc_w, w)) { sg.cache_w = wc_w->packed; sg.status_w = wc_w->status; }
    if (usable(wc_t, target)) { sg.cache_t = wc_t->packed; sg.status_t = wc_t->status; }
    if (ref)
        for (int k = 0; k < n_seg; ++k) sg.w2_off[k] = mlp::wcache_w2_offset(ref, k) - sg.off[k];
    return 0;
}
}  // namespace

extern "C" int mpg_adam_polyak(float* w, float* m, float* v, float* target, const float* grad, const int* seg_sizes,
                               int n_seg, const float* lr_t, const int* do_adam, const int* do_polyak, float tau,
                               const int* skip_flags, int n_skip_flags, const mpg_wcache_t* wc_w,
                               const mpg_wcache_t* wc_target, mpg_stream_t stream) {
    Segs sg;
    int maxn = 0;
    MPG_REQUIRE(w && m && v && grad && lr_t && do_adam && do_polyak &&
                    fill_adam(sg, n_seg, seg_sizes, w, target, lr_t, do_adam, do_polyak, wc_w, wc_target, &maxn) == 0,
                "mpg_adam_polyak: bad argument");
    hipLaunchKernelGGL(k_adam_polyak, dim3((maxn + 255) / 256, n_seg), dim3(256), 0, mpg_stream(stream), sg, w, m, v,
                       target, grad, tau, skip_flags, skip_flags ? n_skip_flags : 0);
    MPG_CHECK_LAUNCH("k_adam_polyak");
    return MPG_OK;
}

extern "C" int mpg_sq_partials(const float* grad, const int* seg_sizes, int n_seg, float* sq_part, mpg_stream_t stream) {
    Segs sg;
    MPG_REQUIRE(grad && sq_part && fill(sg, n_seg, seg_sizes) > 0, "mpg_sq_partials: bad argument");
    hipLaunchKernelGGL(k_sq_blocks, dim3(CLIP_PARTS, n_seg), dim3(256), 0, mpg_stream(stream), sg, grad, sq_part);
    MPG_CHECK_LAUNCH("k_sq_blocks");
    return MPG_OK;
}

extern "C" int mpg_clip_adam_polyak(float* w, float* m, float* v, float* target, float* grad, const float* sq_part,
                                    const int* seg_sizes, int n_seg, float clip, const float* lr_t, const int* do_adam,
                                    const int* do_polyak, float tau, float* norms, int* nonfinite_flags,
                                    const mpg_wcache_t* wc_w, const mpg_wcache_t* wc_target, mpg_stream_t stream) {
    Segs sg;
    int maxn = 0;
    MPG_REQUIRE(w && m && v && grad && sq_part && norms && lr_t && do_adam && do_polyak && clip > 0.f &&
                    fill_adam(sg, n_seg, seg_sizes, w, target, lr_t, do_adam, do_polyak, wc_w, wc_target, &maxn) == 0,
                "mpg_clip_adam_polyak: bad argument");
    hipLaunchKernelGGL(k_clip_adam_polyak, dim3((maxn + 255) / 256, n_seg), dim3(256), 0, mpg_stream(stream), sg, w, m, v,
                       target, grad, sq_part, clip, tau, norms, nonfinite_flags);
    MPG_CHECK_LAUNCH("k_clip_adam_polyak");
    return MPG_OK;
}

namespace {
__global__ void k_sum_slots(const float* __restrict__ slots, int n_slots, size_t stride, int n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = slots[i];
    for (int r = 1; r < n_slots; ++r) s += slots[(size_t)r * stride + i];       // rank order: the same association on every replica
    out[i] = s;
}
// The rank-order sum AND the clip's partial sums of squares of the result in one pass (round 6: an exchanged gradient no longer
// needs a launch of k_sq_blocks behind the exchange).  Block (b, k), k < n_seg, owns exactly the elements k_sq_blocks' block (b, k)
// reads - same per-thread order, same block reduction - so the partials are the same bits; row n_seg of the grid sums the tail behind
// the networks (the statistics) without partials.
__global__ void __launch_bounds__(256) k_sum_slots_sq(const Segs sg, const float* __restrict__ slots, int n_slots, size_t stride, int n,
                                                      float* __restrict__ out, float* __restrict__ part) {
    __shared__ float red[256];
    const int k = blockIdx.y, b = blockIdx.x;
    const bool tail = k == sg.n_seg;
    const int off = tail ? sg.off[sg.n_seg - 1] + sg.n[sg.n_seg - 1] : sg.off[k];
    const int len = tail ? n - off : sg.n[k];
    float a = 0.f;
    for (int i = b * 256 + threadIdx.x; i < len; i += CLIP_PARTS * 256) {
        float s = slots[off + i];
        for (int r = 1; r < n_slots; ++r) s += slots[(size_t)r * stride + off + i];    // rank order: the same association on every replica
        out[off + i] = s;
        a = fmaf(s, s, a);
    }
    if (tail) return;                                   // (block-uniform)
    const float tot = mpg_block_sum256(a, red);
    if (threadIdx.x == 0) part[k * CLIP_PARTS + b] = tot;
}
}  // namespace

extern "C" int mpg_sum_slots_sq(const float* slots, int n_slots, size_t slot_stride, int n, float* out, const int* seg_sizes, int n_seg,
                                float* sq_part, mpg_stream_t stream) {
    Segs sg;
    const int covered = fill(sg, n_seg, seg_sizes);
    MPG_REQUIRE(slots && out && sq_part && n_slots > 0 && n > 0 && slot_stride >= (size_t)n && covered > 0 && covered <= n,
                "mpg_sum_slots_sq: bad argument");
    hipLaunchKernelGGL(k_sum_slots_sq, dim3(CLIP_PARTS, n_seg + (covered < n ? 1 : 0)), dim3(256), 0, mpg_stream(stream), sg, slots, n_slots,
                       slot_stride, n, out, sq_part);
    MPG_CHECK_LAUNCH("k_sum_slots_sq");
    return MPG_OK;
}

extern "C" int mpg_sum_slots(const float* slots, int n_slots, int n, float* out, mpg_stream_t stream) {
    MPG_REQUIRE(slots && out && n_slots > 0 && n > 0, "mpg_sum_slots: bad argument");
    hipLaunchKernelGGL(k_sum_slots, dim3((n + 255) / 256), dim3(256), 0, mpg_stream(stream), slots, n_slots, (size_t)n, n, out);
    MPG_CHECK_LAUNCH("k_sum_slots");
    return MPG_OK;
}

extern "C" int mpg_sum_slots_strided(const float* slots, int n_slots, size_t slot_stride, int n, float* out, mpg_stream_t stream) {
    MPG_REQUIRE(slots && out && n_slots > 0 && n > 0 && slot_stride >= (size_t)n, "mpg_sum_slots_strided: bad argument");
    hipLaunchKernelGGL(k_sum_slots, dim3((n + 255) / 256), dim3(256), 0, mpg_stream(stream), slots, n_slots, slot_stride, n, out);
    MPG_CHECK_LAUNCH("k_sum_slots");
    return MPG_OK;
}
