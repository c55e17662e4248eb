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
