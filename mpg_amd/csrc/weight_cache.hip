// Weight cache: pre-packed register images of the 256x256 hidden kernels.
//
// Every engine kernel starts by loading its wave's slice of W2 (forward) or W2^T (backward) into registers.  From the
// Keras-ordered vector that is 128 strided dword loads per lane (the transposed read touches 16 cache lines per wave
// instruction); at B = 4096 - one 16-row group per CU - this prologue costs more than the kernel's arithmetic.  A cache
// keeps, per network, both images in exactly the order the lanes consume them (1 KiB coalesced float4 loads).
// It is an ACCELERATION ONLY and a CALLER-OWNED object (mpg_wcache_t, include/mpg_hip.h): the library stores nothing;
// launchers resolve the packed image of a network through the descriptors the caller passes with the call, and
// parameter buffers that no descriptor covers take the strided path with identical results
// (tests/test_networks_gpu.py runs both).
#include "mlp_core.h"

namespace {

struct PackArgs {
    const float* base;
    float* cache;
    int n_nets;
    int w2_off[8];
    int w3_n[8];      // entries of the output kernel W3 (256 * out_dim), which follows W2 and b2
    int* status;      // nullable: MPG_STATUS_PARAMETER_RANGE is OR-ed into it
};

// one thread per (net, direction, packed 32-bit word)
__global__ void k_pack(const PackArgs a) {
    const int net = blockIdx.y >> 1, dir = blockIdx.y & 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // 0 .. 65535
    const float* W2 = a.base + a.w2_off[net];
    float* out = a.cache + ((size_t)net * 2 + dir) * (mlp::H * mlp::H);
    // the output kernel W3 has the same envelope (the reverse layer's 16 * out * |W3| operands must stay in fp16 range, mlp_core.h)
    // although it has no packed image: whatever route the parameters arrived by (set_weights, checkpoint restore, a broadcast), the
    // (re)pack is where they are checked - the direction-0 blocks scan it
    if (dir == 0 && idx < a.w3_n[net] && a.status && !(fabsf(W2[mlp::H * mlp::H + mlp::H + idx]) < mlp::P_LIMIT))
        atomicOr(a.status, MPG_STATUS_PARAMETER_RANGE);
#ifdef MPG_SPLIT
    // word (img_slot(wave, v)*64 + lane)*4 + r, v = (kb*2 + t)*2 + part: the packed pair (k0, k0 + 1), k0 = 32 kb + 8 (lane>>4) + 2 r,
    // of output column n = 32 wave + 16 t + (lane&15); part 0 = hi halves, 1 = lo halves of W * W_SCALE (mlp_core.h)
    const int r = idx & 3, lane = (idx >> 2) & 63;
    int wave, v;
    mlp::img_unslot(idx >> 8, wave, v);                          // (the 1 KiB blocks of the eight waves are interleaved: mlp_core.h img_slot)
    const int part = v & 1, t = (v >> 1) & 1, kb = v >> 2;
    const int k0 = 32 * kb + 8 * (lane >> 4) + 2 * r, n = 32 * wave + 16 * t + (lane & 15);
    // a hidden-kernel entry beyond the envelope (|w| >= 1023.5, include/mpg_hip.h) enters clamped and is reported
    const float r0 = dir == 0 ? W2[k0 * mlp::H + n] : W2[n * mlp::H + k0], r1 = dir == 0 ? W2[(k0 + 1) * mlp::H + n] : W2[n * mlp::H + k0 + 1];
    if (a.status && (!(fabsf(r0) < mlp::P_LIMIT) || !(fabsf(r1) < mlp::P_LIMIT))) atomicOr(a.status, MPG_STATUS_PARAMETER_RANGE);
    const float w0 = fminf(fmaxf(r0 * mlp::W_SCALE, -65504.f), 65504.f), w1 = fminf(fmaxf(r1 * mlp::W_SCALE, -65504.f), 65504.f);
    float hi, lo;
    mlp::split_pack2(w0, w1, hi, lo);
    out[idx] = part == 0 ? hi : lo;
#else
    const int e = idx & 3, lane = (idx >> 2) & 63, t = (idx >> 8) & 1, q4 = (idx >> 9) & 15, wave = idx >> 13;
    const int c = lane & 15, rg = lane >> 4;
    const int k = 4 * (4 * q4 + e) + rg;                         // contraction index of the MFMA step
    const int n = 32 * wave + 16 * t + c;                        // output column owned by the lane
    out[idx] = dir == 0 ? W2[k * mlp::H + n] : W2[n * mlp::H + k];
#endif
}

inline bool wc_ok(const mpg_wcache_t* wc) { return wc && wc->params && wc->packed && wc->n_nets > 0 && wc->n_nets <= 8; }

}  // namespace

namespace mlp {

// offset (floats) of network k's W2 inside wc->params
int wcache_w2_offset(const mpg_wcache_t* wc, int k) {
    int off = 0;
    for (int j = 0; j < k; ++j) off += net_size(wc->in_dim[j], wc->out_dim[j]);
    return off + wc->in_dim[k] * H + H;
}

const float* wcache_lookup(const mpg_wcache_t* wc, const float* W2, int dir) {
    if (!wc_ok(wc) || !W2) return nullptr;
    int off = 0;
    for (int k = 0; k < wc->n_nets; ++k) {
        if (wc->params + off + wc->in_dim[k] * H + H == W2) return wc->packed + ((size_t)k * 2 + dir) * (H * H);
        off += net_size(wc->in_dim[k], wc->out_dim[k]);
    }
    return nullptr;
}

const float* weight_cache_lookup(const mpg_cfg_t* cfg, const float* W2, int dir) {
    if (!cfg) return nullptr;
    for (int i = 0; i < 2; ++i)
        if (const float* p = wcache_lookup(cfg->wcache[i], W2, dir)) return p;
    return nullptr;
}

}  // namespace mlp

extern "C" size_t mpg_weight_cache_floats(int n_nets) { return n_nets > 0 ? (size_t)n_nets * 2 * mlp::H * mlp::H : 0; }

extern "C" int mpg_weight_cache_pack(const mpg_wcache_t* wc, mpg_stream_t stream) {
    MPG_REQUIRE(wc_ok(wc), "mpg_weight_cache_pack: incomplete descriptor");
    PackArgs a;
    a.base = wc->params; a.cache = wc->packed; a.n_nets = wc->n_nets; a.status = wc->status;
    for (int k = 0; k < 8; ++k) {
        a.w2_off[k] = k < wc->n_nets ? mlp::wcache_w2_offset(wc, k) : 0;
        a.w3_n[k] = k < wc->n_nets ? mlp::H * wc->out_dim[k] : 0;
    }
    hipLaunchKernelGGL(k_pack, dim3(mlp::H * mlp::H / 256, 2 * wc->n_nets), dim3(256), 0, mpg_stream(stream), a);
    MPG_CHECK_LAUNCH("k_pack");
    return MPG_OK;
}
