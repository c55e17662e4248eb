// Weight cache: pre-packed register images of the 256x256 hidden kernels.
//
// Every engine kernel starts by loading its wave's slice of W2 (forward) or W2^T (backward) into registers.  From the
// Keras-ordered vector that is 128 strided dword loads per lane (the transposed read touches 16 cache lines per wave
// instruction); at B = 4096 - one 16-row group per CU - this prologue costs more than the kernel's arithmetic.  The
// cache keeps, per network, both images in exactly the order the lanes consume them (1 KiB coalesced float4 loads).
// It is an ACCELERATION ONLY: unbound parameter buffers take the strided path with identical results
// (tests/test_networks_gpu.py runs both).  Whoever writes a bound buffer must refresh it; mpg_adam_polyak does so itself.
#include <mutex>
#include <vector>

#include "mlp_core.h"

namespace {

struct Binding {
    const float* base;
    size_t n;
    int n_nets;
    int off[8], in_dim[8], out_dim[8];
    float* cache;
};
std::vector<Binding> g_bind;
std::mutex g_mu;

struct PackArgs {
    const float* base;
    float* cache;
    int n_nets;
    int w2_off[8];
};

// one thread per (net, direction, packed element)
__global__ void k_pack(const PackArgs a) {
    const int net = blockIdx.y >> 1, dir = blockIdx.y & 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // 0 .. 65535
    const int e = idx & 3, lane = (idx >> 2) & 63, t = (idx >> 8) & 1, q4 = (idx >> 9) & 15, wave = idx >> 13;
    const int c = lane & 15, rg = lane >> 4;
    const int k = 4 * (4 * q4 + e) + rg;                         // contraction index of the MFMA step
    const int n = 32 * wave + 16 * t + c;                        // output column owned by the lane
    const float* W2 = a.base + a.w2_off[net];
    a.cache[((size_t)net * 2 + dir) * (mlp::H * mlp::H) + idx] = dir == 0 ? W2[k * mlp::H + n] : W2[n * mlp::H + k];
}

int refresh(const Binding& b, hipStream_t s) {
    PackArgs a;
    a.base = b.base; a.cache = b.cache; a.n_nets = b.n_nets;
    for (int k = 0; k < 8; ++k) a.w2_off[k] = k < b.n_nets ? b.off[k] + b.in_dim[k] * mlp::H + mlp::H : 0;
    hipLaunchKernelGGL(k_pack, dim3(mlp::H * mlp::H / 256, 2 * b.n_nets), dim3(256), 0, s, a);
    MPG_CHECK_LAUNCH("k_pack");
    return MPG_OK;
}

}  // namespace

namespace mlp {
const float* weight_cache_lookup(const float* W2, int dir) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (const Binding& b : g_bind) {
        if (W2 < b.base || W2 >= b.base + b.n) continue;
        for (int k = 0; k < b.n_nets; ++k)
            if (b.base + b.off[k] + b.in_dim[k] * H + H == W2) return b.cache + ((size_t)k * 2 + dir) * (H * H);
    }
    return nullptr;
}
}  // namespace mlp

// refresh if `params` is the base of a bound buffer; no-op (MPG_OK) otherwise.  Used by mpg_adam_polyak.
int weight_cache_refresh_if_bound(const float* params, hipStream_t s) {
    Binding b;
    bool found = false;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (const Binding& x : g_bind)
            if (x.base == params) { b = x; found = true; break; }
    }
    return found ? refresh(b, s) : MPG_OK;
}

// binding of `params` (base pointer) if any: cache pointer and the offset of every network's W2 inside params
bool weight_cache_info(const float* params, float** cache, int* w2_off, int* n_nets) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (const Binding& b : g_bind)
        if (b.base == params) {
            *cache = b.cache;
            *n_nets = b.n_nets;
            for (int k = 0; k < b.n_nets; ++k) w2_off[k] = b.off[k] + b.in_dim[k] * mlp::H + mlp::H;
            return true;
        }
    return false;
}

extern "C" size_t mpg_weight_cache_floats(int n_nets) { return n_nets > 0 ? (size_t)n_nets * 2 * mlp::H * mlp::H : 0; }

extern "C" int mpg_weight_cache_bind(const float* params, const int* in_dims, const int* out_dims, int n_nets, float* cache,
                                     mpg_stream_t stream) {
    MPG_REQUIRE(params && in_dims && out_dims && cache && n_nets > 0 && n_nets <= 8, "mpg_weight_cache_bind: bad argument");
    Binding b;
    b.base = params; b.n_nets = n_nets; b.cache = cache;
    int off = 0;
    for (int k = 0; k < n_nets; ++k) {
        b.off[k] = off; b.in_dim[k] = in_dims[k]; b.out_dim[k] = out_dims[k];
        off += mlp::net_size(in_dims[k], out_dims[k]);
    }
    b.n = off;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (size_t i = 0; i < g_bind.size(); ++i)
            if (g_bind[i].base == params) { g_bind.erase(g_bind.begin() + i); break; }
        g_bind.push_back(b);
    }
    return refresh(b, mpg_stream(stream));
}

extern "C" int mpg_weight_cache_refresh(const float* params, mpg_stream_t stream) {
    MPG_REQUIRE(params, "mpg_weight_cache_refresh: null pointer");
    return weight_cache_refresh_if_bound(params, mpg_stream(stream));
}

extern "C" int mpg_weight_cache_unbind(const float* params) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = 0; i < g_bind.size(); ++i)
        if (g_bind[i].base == params) { g_bind.erase(g_bind.begin() + i); break; }
    return MPG_OK;
}
