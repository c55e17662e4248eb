// K9 (part 2): on-device prioritized replay - sum / min segment trees, proportional sampling, IS weights.
// Reference: utils/segment_tree.py:13-151 (SegmentTree, SumSegmentTree.find_prefixsum_idx, MinSegmentTree) and
// buffer.py:94-189 (PrioritizedReplayBuffer).  The trees keep the reference's heap layout (node 1 = root, leaves at
// [capacity, 2*capacity)) and its float64 values (python floats), every internal node being left (+|min) right, so the
// tree CONTENT is bit-identical to the reference's whatever the update order; sampling indices are therefore exact.
//
// Batched update = (1) set the leaves, last occurrence of a duplicate index wins like the sequential python loop,
// (2) rebuild the internal nodes: bottom 10 levels per 1024-leaf subtree inside one workgroup, top levels by one
// workgroup.  No float atomics.
#include "replay_common.h"

namespace {

constexpr int SUB = 1024;   // leaves per subtree handled by one workgroup

__global__ void k_stamp(int n, const int* __restrict__ idx, int* __restrict__ stamp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicMax(&stamp[idx[i]], i);
}

// const_prio (nullable): every entry takes THIS priority instead of prio[i] - the max-priority leaves of freshly added transitions
// (buffer.py:133-136), read as the float64 it is kept in (the reference's _max_priority is a python float)
__global__ void k_set_leaves(int n, int capacity, const int* __restrict__ idx, const float* __restrict__ prio,
                             double alpha, double eps, int* __restrict__ stamp, double* __restrict__ sum_tree,
                             double* __restrict__ min_tree, const double* __restrict__ const_prio) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int leaf = idx[i];
    if (stamp[leaf] != i) return;                      // a later entry of the batch overrides this one
    const double p = const_prio ? *const_prio : fabs((double)prio[i]) + eps;      // canonical PER: |td| + eps (the shipped ctor is dead code, SURVEY B-3)
    const double v = pow(p, alpha);                    // buffer.py:185-187
    sum_tree[capacity + leaf] = v;
    min_tree[capacity + leaf] = v;
}
// the same with ONE atomic per wave for the running maximum (65 536 atomics on one address were 12 of the kernel's 17.6 us at TD3's batch):
// every lane takes part in the wave's maximum (lanes without a leaf of their own contribute 0: every p is > 0)
__global__ void __launch_bounds__(256) k_set_leaves_wmax(int n, int capacity, const int* __restrict__ idx, const float* __restrict__ prio,
                                                         double alpha, double eps, const int* __restrict__ stamp, double* __restrict__ sum_tree,
                                                         double* __restrict__ min_tree, double* __restrict__ max_prio) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double mine = 0.0;
    if (i < n) {
        const int leaf = idx[i];
        const float pr = prio[i];                        // (requested with the index: behind the stamp test it was a third round trip)
        if (stamp[leaf] == i) {                          // (a later entry of the batch overrides the others)
            const double p = fabs((double)pr) + eps;
            const double v = pow(p, alpha);
            sum_tree[capacity + leaf] = v;
            min_tree[capacity + leaf] = v;
            mine = p;
        }
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) mine = fmax(mine, __shfl_xor(mine, m, 64));
    // (the running maximum only grows: a wave whose maximum does not exceed what it reads there has nothing to add - a stale read costs
    // one redundant atomic, never a lost one; 1024 atomics on one address are ~35 ns each).  Float64 like the reference's python float
    // (buffer.py:189); p > 0, so the bit pattern orders like the value
    if ((threadIdx.x & 63) == 0 && mine > 0.0 && mine > *reinterpret_cast<volatile double*>(max_prio))
        atomicMax(reinterpret_cast<unsigned long long*>(max_prio), (unsigned long long)__double_as_longlong(mine));
}

__global__ void k_unstamp(int n, const int* __restrict__ idx, int* __restrict__ stamp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) stamp[idx[i]] = -1;
}

// rebuild nodes of one 1024-leaf subtree (levels below the subtree root), SegmentTree.__setitem__ :90-97 semantics
// n_unstamp > 0: the batch's stamps are also cleared here (k_unstamp's work, one launch less: the stamps are only read by
// k_set_leaves, which has finished)
__global__ void __launch_bounds__(512) k_rebuild_bottom(int capacity, double* __restrict__ sum_tree, double* __restrict__ min_tree, int n_unstamp,
                                                        const int* __restrict__ idx, int* __restrict__ stamp) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_unstamp; i += gridDim.x * blockDim.x) stamp[idx[i]] = -1;
    const int nsub_leaves = capacity < SUB ? capacity : SUB;
    const int sub = blockIdx.x;                         // subtree index
    // node ids of level with `w` nodes inside this subtree: first = (capacity / nsub_leaves * w') ... compute per level
    for (int w = nsub_leaves / 2; w >= 1; w >>= 1) {    // w = nodes of this subtree on the level being written
        const int level_first = (capacity / nsub_leaves) * w;        // first node id of that level in the whole tree
        for (int j = threadIdx.x; j < w; j += blockDim.x) {
            const int node = level_first + sub * w + j;
            sum_tree[node] = sum_tree[2 * node] + sum_tree[2 * node + 1];
            min_tree[node] = fmin(min_tree[2 * node], min_tree[2 * node + 1]);
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(512) k_rebuild_top(int ntop, double* __restrict__ sum_tree, double* __restrict__ min_tree) {
    // ntop = number of subtree roots (a power of two); rebuild levels above them
    for (int w = ntop / 2; w >= 1; w >>= 1) {
        for (int j = threadIdx.x; j < w; j += blockDim.x) {
            const int node = w + j;
            sum_tree[node] = sum_tree[2 * node] + sum_tree[2 * node + 1];
            min_tree[node] = fmin(min_tree[2 * node], min_tree[2 * node + 1]);
        }
        __syncthreads();
    }
}

// Touched-path update (batch << capacity): only the ancestors of the n updated leaves are recomputed, level by level, by ONE workgroup
// (a level's nodes read the level below, written by other threads of the same workgroup: block-wide barrier + fence per level).
// Several leaves under one ancestor recompute the same left (+|min) right from the same children - identical values, so the duplicate
// stores are harmless and the tree content stays bit-identical to the full rebuild's (and to SegmentTree.__setitem__'s, :90-97).
__global__ void __launch_bounds__(1024) k_update_paths(int capacity, int n, const int* __restrict__ idx, double* __restrict__ sum_tree,
                                                        double* __restrict__ min_tree) {
    for (int node_shift = 1; (capacity >> node_shift) >= 1; ++node_shift) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int node = (capacity + idx[i]) >> node_shift;
            sum_tree[node] = sum_tree[2 * node] + sum_tree[2 * node + 1];
            min_tree[node] = fmin(min_tree[2 * node], min_tree[2 * node + 1]);
        }
        __threadfence_block();
        __syncthreads();
    }
}

// GATHER: the sampled transition is also copied out of the ring by the thread that found it (ReplayBuffer._encode_sample,
// buffer.py:57-68, 161-164): one launch instead of k_sample + k_gather (11 + 11 us at 65 536 rows; the row's random reads are
// issued right behind the descent instead of in a launch of their own)
struct GatherOut {
    Ring ring;
    int od, ad;
    float *obs, *act, *rew, *obs2, *done;
};
template <bool GATHER>
__global__ void k_sample(int capacity, int n_storage, int n, const double* __restrict__ sum_tree,
                         const double* __restrict__ min_tree, const double* __restrict__ u, uint32_t k0, uint32_t k1,
                         uint32_t c1, uint32_t c2, double beta, int* __restrict__ idx, float* __restrict__ is_w, GatherOut go) {
    // the top levels of the sum tree (nodes 1 .. 2047: 16 KB) are staged in LDS once per workgroup: the first 10 of the 19 dependent
    // reads of a descent at capacity 2^19 then cost an LDS access instead of an L2 round trip (round 3: 41 us for sample + gather at
    // B = 65 536).  Same values, same comparisons: the indices are bit-identical.
    constexpr int TOPN = 2048;
    __shared__ double sTop[TOPN];
    const int staged = 2 * capacity < TOPN ? 2 * capacity : TOPN;
    for (int j = threadIdx.x; j < staged; j += blockDim.x) sTop[j] = sum_tree[j];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double ui;
    if (u) {
        ui = u[i];
    } else {
        const Philox4 p = philox4x32_10((uint32_t)i, c1, c2, 0x9e4u, k0, k1);
        // 52 random bits + 0.5: exactly representable, so ui lies STRICTLY inside (0, 1) (with 53 bits the largest value + 0.5 rounds to
        // 2^53: ui == 1, prefix == total, and the descent ends in the rightmost - unfilled - leaf)
        ui = ((double)(((uint64_t)p.v[0] << 20) ^ (uint64_t)(p.v[1] >> 12)) + 0.5) * (1.0 / 4503599627370496.0);
    }
    const double total = sTop[1];                        // == sum(0, len(storage)) bit for bit (unfilled leaves are exact zeros)
    double prefix = ui * total;                          // buffer.py:141
    int node = 1;
    while (node < capacity && 2 * node + 1 < staged) {   // find_prefixsum_idx, segment_tree.py:133-140: the staged levels ...
        const double left = sTop[2 * node];
        if (left > prefix) {
            node = 2 * node;
        } else {
            prefix -= left;
            node = 2 * node + 1;
        }
    }
    while (node < capacity) {                            // ... and the rest of the way down
        const double left = sum_tree[2 * node];
        if (left > prefix) {
            node = 2 * node;
        } else {
            prefix -= left;
            node = 2 * node + 1;
        }
    }
    // A caller-supplied u keeps find_prefixsum_idx's own answer (u = 1 -> the rightmost leaf, segment_tree.py:133-140: the reference
    // would then fail on its storage list).  The library's own draws and every gathering launch never leave the filled slots:
    // ui * total can still round up to `total` once in 2^52 draws.
    const int leaf = (u && !GATHER) ? node - capacity : min(node - capacity, n_storage - 1);
    idx[i] = leaf;
    if constexpr (GATHER) gather_row(go.ring, leaf, i, go.od, go.ad, go.obs, go.act, go.rew, go.obs2, go.done);
    if (is_w) {                                          // buffer.py:146-158
        const double p_min = min_tree[1] / total;
        const double max_w = pow(p_min * (double)n_storage, -beta);
        const double p_s = sum_tree[capacity + leaf] / total;
        is_w[i] = (float)(pow(p_s * (double)n_storage, -beta) / max_w);
    }
}

__global__ void k_tree_init(int capacity, double* __restrict__ sum_tree, double* __restrict__ min_tree, int* __restrict__ stamp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * capacity) {
        sum_tree[i] = 0.0;                               // neutral elements, segment_tree.py:106,146
        min_tree[i] = __builtin_huge_val();
    }
    if (i < capacity) stamp[i] = -1;
}

int rebuild(int capacity, double* sum_tree, double* min_tree, hipStream_t s, int n_unstamp = 0, const int* idx = nullptr, int* stamp = nullptr) {
    const int nsub = capacity < SUB ? 1 : capacity / SUB;
    hipLaunchKernelGGL(k_rebuild_bottom, dim3(nsub), dim3(512), 0, s, capacity, sum_tree, min_tree, n_unstamp, idx, stamp);
    MPG_CHECK_LAUNCH("k_rebuild_bottom");
    if (nsub > 1) {
        hipLaunchKernelGGL(k_rebuild_top, dim3(1), dim3(512), 0, s, nsub, sum_tree, min_tree);
        MPG_CHECK_LAUNCH("k_rebuild_top");
    }
    return MPG_OK;
}

inline bool pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

// idx[i] = (start + i) % ring_capacity: the slots of freshly added transitions (buffer.py:127-136; their leaves take *max_priority)
__global__ void k_per_add_fill(int n, int start, int ring_capacity, int* __restrict__ idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    idx[i] = (start + i) % ring_capacity;
}

}  // namespace

extern "C" int mpg_per_init(double* sum_tree, double* min_tree, int* stamp, int capacity, mpg_stream_t stream) {
    MPG_REQUIRE(sum_tree && min_tree && stamp && pow2(capacity), "mpg_per_init: capacity must be a power of two (segment_tree.py:40)");
    hipLaunchKernelGGL(k_tree_init, dim3((2 * capacity + 255) / 256), dim3(256), 0, mpg_stream(stream), capacity, sum_tree,
                       min_tree, stamp);
    MPG_CHECK_LAUNCH("k_tree_init");
    return MPG_OK;
}

namespace {
int per_update_impl(double* sum_tree, double* min_tree, int* stamp, int capacity, int n, const int* idx, const float* prio, double alpha,
                    double eps, double* max_priority, const double* const_prio, mpg_stream_t stream);
}
extern "C" int mpg_per_update(double* sum_tree, double* min_tree, int* stamp, int capacity, int n, const int* idx,
                              const float* prio, double alpha, double eps, double* max_priority, mpg_stream_t stream) {
    MPG_REQUIRE(prio, "mpg_per_update: bad argument");
    return per_update_impl(sum_tree, min_tree, stamp, capacity, n, idx, prio, alpha, eps, max_priority, nullptr, stream);
}
namespace {
int per_update_impl(double* sum_tree, double* min_tree, int* stamp, int capacity, int n, const int* idx, const float* prio, double alpha,
                    double eps, double* max_priority, const double* const_prio, mpg_stream_t stream) {
    MPG_REQUIRE(sum_tree && min_tree && stamp && idx && pow2(capacity) && n > 0, "mpg_per_update: bad argument");
    hipStream_t s = mpg_stream(stream);
    const dim3 g((n + 255) / 256), b(256);
    const bool paths = (long)n * 64 <= (long)capacity;
    hipLaunchKernelGGL(k_stamp, g, b, 0, s, n, idx, stamp);
    if (max_priority) hipLaunchKernelGGL(k_set_leaves_wmax, g, b, 0, s, n, capacity, idx, prio, alpha, eps, stamp, sum_tree, min_tree, max_priority);
    else hipLaunchKernelGGL(k_set_leaves, g, b, 0, s, n, capacity, idx, prio, alpha, eps, stamp, sum_tree, min_tree, const_prio);
    if (paths) hipLaunchKernelGGL(k_unstamp, g, b, 0, s, n, idx, stamp);
    MPG_CHECK_LAUNCH("mpg_per_update");
    // n log2(capacity) node updates by one workgroup against 2 * capacity by the whole chip: the paths win for small batches (B = 256
    // into 2^19 leaves: 19 levels of 256 nodes)
    if (paths) {
        hipLaunchKernelGGL(k_update_paths, dim3(1), dim3(1024), 0, s, capacity, n, idx, sum_tree, min_tree);
        MPG_CHECK_LAUNCH("k_update_paths");
        return MPG_OK;
    }
    return rebuild(capacity, sum_tree, min_tree, s, n, idx, stamp);        // (clears the stamps on its way)
}
}  // namespace

extern "C" int mpg_per_sample(const double* sum_tree, const double* min_tree, int capacity, int n_storage, int n,
                              const double* u, uint64_t seed, uint64_t ctr, double beta, int* idx, float* is_weight,
                              mpg_stream_t stream) {
    MPG_REQUIRE(sum_tree && min_tree && idx && pow2(capacity) && n > 0 && n_storage > 0 && n_storage <= capacity,
                "mpg_per_sample: bad argument");
    hipLaunchKernelGGL(k_sample<false>, dim3((n + 255) / 256), dim3(256), 0, mpg_stream(stream), capacity, n_storage, n, sum_tree,
                       min_tree, u, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)ctr, (uint32_t)(ctr >> 32), beta,
                       idx, is_weight, GatherOut{});
    MPG_CHECK_LAUNCH("k_sample");
    return MPG_OK;
}

extern "C" int mpg_per_sample_gather(const double* sum_tree, const double* min_tree, int capacity, int n_storage, int n,
                                     const double* u, uint64_t seed, uint64_t ctr, double beta, int* idx, float* is_weight,
                                     int obs_dim, int act_dim, const float* ring_obs, const float* ring_act, const float* ring_rew,
                                     const float* ring_obs2, const uint8_t* ring_done, float* o_obs, float* o_act, float* o_rew,
                                     float* o_obs2, float* o_done, mpg_stream_t stream) {
    MPG_REQUIRE(sum_tree && min_tree && idx && pow2(capacity) && n > 0 && n_storage > 0 && n_storage <= capacity,
                "mpg_per_sample_gather: bad argument");
    MPG_REQUIRE(ring_obs && ring_act && ring_rew && ring_obs2 && ring_done && o_obs && o_act && o_rew && o_obs2 && obs_dim > 0 &&
                    obs_dim <= MAXOD && act_dim > 0 && act_dim <= MAXAD,
                "mpg_per_sample_gather: bad ring / output argument");
    GatherOut go;
    go.ring = Ring{const_cast<float*>(ring_obs), const_cast<float*>(ring_act), const_cast<float*>(ring_rew), const_cast<float*>(ring_obs2),
                   const_cast<uint8_t*>(ring_done)};
    go.od = obs_dim; go.ad = act_dim; go.obs = o_obs; go.act = o_act; go.rew = o_rew; go.obs2 = o_obs2; go.done = o_done;
    hipLaunchKernelGGL(k_sample<true>, dim3((n + 255) / 256), dim3(256), 0, mpg_stream(stream), capacity, n_storage, n, sum_tree,
                       min_tree, u, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)ctr, (uint32_t)(ctr >> 32), beta,
                       idx, is_weight, go);
    MPG_CHECK_LAUNCH("k_sample (gather)");
    return MPG_OK;
}

extern "C" int mpg_per_add(double* sum_tree, double* min_tree, int* stamp, int capacity, int ring_capacity, int start, int n,
                           double alpha, double* max_priority, int* idx_scratch, mpg_stream_t stream) {
    MPG_REQUIRE(sum_tree && min_tree && stamp && max_priority && idx_scratch && n > 0 && ring_capacity > 0 &&
                    ring_capacity <= capacity && start >= 0 && start < ring_capacity,
                "mpg_per_add: bad argument");
    hipLaunchKernelGGL(k_per_add_fill, dim3((n + 255) / 256), dim3(256), 0, mpg_stream(stream), n, start, ring_capacity, idx_scratch);
    MPG_CHECK_LAUNCH("k_per_add_fill");
    return per_update_impl(sum_tree, min_tree, stamp, capacity, n, idx_scratch, nullptr, alpha, 0.0, nullptr, max_priority, stream);
}

