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
