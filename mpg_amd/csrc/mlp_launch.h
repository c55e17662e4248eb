// Host-side launchers of the generic network kernels (internal to libmpg_hip.so; not part of the C ABI).
#pragma once
#include "mlp_core.h"

namespace mlp {

struct XSpec {              // network input = [x0 (d0 columns, column i < n_scaled multiplied by scale[i]) | x1 (d1 columns)]
    const float* x0;
    int d0, ld0;            // ld = row stride in floats
    const float* x1;
    int d1, ld1;
    float scale[8];
    int n_scaled;
};

inline XSpec xspec(const float* x0, int d0, const float* x1, int d1, const float* scale, int n_scaled) {
    XSpec s;
    s.x0 = x0; s.d0 = d0; s.ld0 = d0; s.x1 = x1; s.d1 = d1; s.ld1 = d1; s.n_scaled = n_scaled;
    for (int i = 0; i < 8; ++i) s.scale[i] = (scale && i < n_scaled) ? scale[i] : 1.f;
    return s;
}

struct OutSpec {            // y = out_scale * tanh(z) (out_tanh) or z; optional Philox N(0, sigma) exploration noise
    int out_tanh;
    float out_scale;
    float sigma;
    uint64_t seed, ctr;
};

// y[rows][ldy] (first `ou` columns) = net(x); optional G16 stashes of both hidden activations.
int launch_forward(const float* params, int in_dim, int out_dim, int ou, int rows, const XSpec& x, const OutSpec& o,
                   float* y, int ldy, float* h1, float* h2, hipStream_t s);

// Input-side backward: dy [rows][lddy] is dL/d(output after activation); yout [rows][ldyo] the forward outputs.
// Writes (each nullable) dz1, dz2 (G16), dz3 [rows][ou] and dx [rows][lddx] (in_dim columns, w.r.t. the network
// input as seen by the first layer, i.e. after scaling).
int launch_backward(const float* params, int in_dim, int out_dim, int ou, int rows, const float* dy, int lddy,
                    const float* yout, int ldyo, int out_tanh, float out_scale, const float* h1, const float* h2,
                    float* dz1, float* dz2, float* dz3, float* dx, int lddx, hipStream_t s);

// Weight gradient of one network over `rows` rows from stashes; result (net_size floats, fully reduced over rows,
// accumulate == 0: overwritten) in grad.  ws must hold wgrad_workspace_floats(rows, in_dim, out_dim) floats.
size_t wgrad_workspace_floats(int rows, int in_dim, int out_dim);
int launch_wgrad(int in_dim, int out_dim, int ou, int rows, const XSpec& x, const float* h1, const float* h2,
                 const float* dz1, const float* dz2, const float* dz3, float* grad, float* ws, hipStream_t s);

inline size_t stash_floats(int rows) { return (size_t)((rows + GROUP - 1) / GROUP) * GROUP * H; }

}  // namespace mlp
