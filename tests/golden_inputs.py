"""Seeded synthetic inputs shared by tests and bench (same laws as tests/golden/make_golden.py)."""
import numpy as np


def orthogonal(rng, rows, cols, gain):
    a = rng.standard_normal((max(rows, cols), min(rows, cols)))
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diag(r))
    if rows < cols:
        q = q.T
    return (gain * q[:rows, :cols]).astype(np.float32)


def mlp_weights_flat(rng, din, dout, H=256, bias_jitter=0.05):
    """Orthogonal(sqrt 2 / 1) kernels like model.py:23-36, flat Keras order; small non-zero biases."""
    ws = [orthogonal(rng, din, H, np.sqrt(2.)), (bias_jitter * rng.standard_normal(H)).astype(np.float32),
          orthogonal(rng, H, H, np.sqrt(2.)), (bias_jitter * rng.standard_normal(H)).astype(np.float32),
          orthogonal(rng, H, dout, 1.), (bias_jitter * rng.standard_normal(dout)).astype(np.float32)]
    return np.concatenate([w.ravel() for w in ws]).astype(np.float32)


def reset_law_obs(rng, n):
    """obs ~ PathTrackingEnv.reset() law, path_tracking_env.py:426-437."""
    x = rng.uniform(0, 600, n).astype(np.float32)
    dy = rng.normal(0, 1, n).astype(np.float32)
    dphi = rng.normal(0, np.pi / 9, n).astype(np.float32)
    vx = rng.uniform(15, 25, n).astype(np.float32)
    beta = rng.normal(0, 0.15, n).astype(np.float32)
    vy = (vx * np.tan(beta)).astype(np.float32)
    r = rng.normal(0, 0.3, n).astype(np.float32)
    return np.stack([vx - np.float32(20.), vy, r, dy, dphi, x], 1).astype(np.float32)


# ---- bench-size cases (BASELINE.json configs C2 / C3 / C4): every input is a seeded draw, so a fixture only has to carry
# ---- what the reference computed; tests/golden/make_golden.py and the GPU tests both call these ----------------------
def mlp_weights_list(rng, din, H, dout, bias_jitter=0.05):
    """Keras-shaped list [W1,b1,W2,b2,W3,b3] - the same draws, in the same order, as mlp_weights_flat."""
    return [orthogonal(rng, din, H, np.sqrt(2.)), (bias_jitter * rng.standard_normal(H)).astype(np.float32),
            orthogonal(rng, H, H, np.sqrt(2.)), (bias_jitter * rng.standard_normal(H)).astype(np.float32),
            orthogonal(rng, H, dout, 1.), (bias_jitter * rng.standard_normal(dout)).astype(np.float32)]


BENCH_CASES = {            # name -> (learner, B, seed)
    'c2_mpg_v2_B4096': ('MPG-v2', 4096, 101),
    'c3_nadp_B8192': ('NADP', 8192, 102),
    'c4_td3_B65536': ('TD3', 65536, 103),
}


def bench_case_inputs(name):
    """Seeded inputs of one bench-size case: network weights (Keras lists), a synthetic replay batch and the noise.
    PathTracking batches are NOT physically consistent transitions (obs' is an independent reset-law draw, the raw reward a
    uniform draw in the env's range): the learners treat a batch as data, and this keeps every input a pure function of
    the seed on both sides."""
    kind, B, seed = BENCH_CASES[name]
    rng = np.random.Generator(np.random.PCG64(seed))
    H = 256
    d = {'kind': kind, 'B': B}
    if kind == 'NADP':
        d['nets'] = {'policy': mlp_weights_list(rng, 4, H, 2), 'Q1': mlp_weights_list(rng, 5, H, 1)}
        obs = (rng.standard_normal((B, 4)) * np.array([0.5, 0.1, 0.5, 0.5])).astype(np.float32)
        act = rng.uniform(-3, 3, (B, 1)).astype(np.float32)
        d['batch'] = [obs, act, np.zeros(B, np.float32), obs.copy(), np.zeros(B, np.float32)]
        d['eps_q'] = rng.standard_normal((25, B)).astype(np.float32)
        d['eps_pi'] = rng.standard_normal((25, B)).astype(np.float32)
        return d
    d['nets'] = {'policy': mlp_weights_list(rng, 6, H, 4), 'Q1': mlp_weights_list(rng, 8, H, 1),
                 'Q2': mlp_weights_list(rng, 8, H, 1)}
    obs = reset_law_obs(rng, B)
    act = np.clip(rng.uniform(-1, 1, (B, 2)) + 0.1 * rng.standard_normal((B, 2)), -1.2, 1.2).astype(np.float32)
    rew = rng.uniform(-30., 0., B).astype(np.float32)
    obs2 = reset_law_obs(rng, B)
    d['batch'] = [obs, act, rew, obs2, np.ones(B, np.float32)]
    if kind == 'MPG-v2':
        d['eps'] = rng.standard_normal((25, B)).astype(np.float32)
    else:
        d['smooth_eps'] = rng.standard_normal((B, 2)).astype(np.float32)
    return d


# ---- round 6: the reference's OWN loop code (policy.py:123-171, worker.py:91-119, optimizer.py:330-362) run unmodified ----
# Initial weights and gradient lists are seeded draws regenerated on both sides; the fixtures hold what the reference computed.
LOOP_SEED = 3                 # `args.seed` of the loop cases: worker / env stream 3 * 1000003, replay stream 3 * 7919, learner stream 3 + 12345
NET_DIMS = {                  # case -> ordered (name, in, out) like PolicyWithQs.models (policy.py:72-86)
    'v2': [('Q1', 8, 1), ('Q2', 8, 1), ('policy', 6, 4)],
    'v1': [('Q1', 8, 1), ('policy', 6, 4)],
    'nadp': [('Q1', 5, 1), ('policy', 4, 2)],
    'v2k3': [('Q1', 11, 1), ('Q2', 11, 1), ('policy', 9, 4)],      # MPG-v2 with num_future_data = 3 (obs 9 wide)
}


def loop_case_weights(case, seed=600):
    """Initial ONLINE networks of a loop / apply_gradients case as {name: flat float32}: Orthogonal(sqrt 2 / 1) kernels, ZERO biases
    (model.py:23-36 - what the reference and the device both start from)."""
    rng = np.random.Generator(np.random.PCG64(seed + sorted(NET_DIMS).index(case)))
    return {name: mlp_weights_flat(rng, din, dout, bias_jitter=0.0) for name, din, dout in NET_DIMS[case]}


def apply_case_grads(case, n_iter=6, seed=610):
    """n_iter gradient lists for PolicyWithQs.apply_gradients: per network a flat float32 vector; magnitudes spread over four
    decades so that Adam's normalisation meets small and large second moments."""
    rng = np.random.Generator(np.random.PCG64(seed + sorted(NET_DIMS).index(case)))
    out = []
    for _ in range(n_iter):
        g = {}
        for name, din, dout in NET_DIMS[case]:
            n = (din + 1) * 256 + 257 * 256 + 257 * dout
            g[name] = (rng.standard_normal(n) * 10.0 ** rng.uniform(-5, -1, n)).astype(np.float32)
        out.append(g)
    return out


def split_keras(flat, din, dout, H=256):
    """flat float32 -> Keras-shaped list [W1, b1, W2, b2, W3, b3]"""
    out, o = [], 0
    for shp in [(din, H), (H,), (H, H), (H,), (H, dout), (dout,)]:
        n = int(np.prod(shp))
        out.append(np.asarray(flat[o:o + n], np.float32).reshape(shp))
        o += n
    assert o == len(flat)
    return out
