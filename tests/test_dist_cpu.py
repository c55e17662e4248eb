"""World-size-2 `gloo` coverage of the data-parallel form of the gradient step, on CPU.

Each rank computes SUM-reduced, UN-clipped gradient partials of its shard already scaled by 1/B_global (here with the
oracle standing in for the HIP kernels, which need a GPU), the flat [grads | stats] buffer is all-reduced ONCE through
mpg_amd.dist, and only then clipped per network.  The result must equal the single-process gradient on the full batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import mpg_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _partials(cfg, flat, batch, eps, w, inv_b_global):
    """un-clipped, sum-reduced partials scaled by 1/B_global for one shard: [q1 | q2 | policy | q_loss1 q_loss2 ret0 ret25]"""
    nets = O.Nets(cfg, flat, target_scale=0.97, dtype=torch.float64)
    obs, act, rew, obs1 = [torch.as_tensor(b).double() for b in batch[:4]]
    y = O.clipped_double_q_target(cfg, nets, rew, obs1)
    po = O.process_obses(cfg, obs)
    out, stats = [], []
    for nm in ('Q1', 'Q2'):
        loss = 0.5 * torch.sum((nets.q(nm, po, act) - y) ** 2) * inv_b_global
        out += [g.reshape(-1) for g in torch.autograd.grad(loss, nets.w[nm])]
        stats.append(loss.detach())
    reduced, _, allret = O.model_rollout_for_policy_update(cfg, nets, obs, torch.as_tensor(eps).double())
    sums = allret.sum(1)                                   # sum over this shard's rows, per slice
    loss = -(w[0] * sums[0] + w[1] * sums[25]) * inv_b_global
    out += [g.reshape(-1) for g in torch.autograd.grad(loss, nets.w['policy'])]
    stats += [sums[0].detach(), sums[25].detach()]
    return torch.cat(out + [torch.stack(stats)])


def _inputs():
    from tests.golden_inputs import mlp_weights_flat, reset_law_obs
    rng = np.random.Generator(np.random.PCG64(42))
    B = 32
    flat = {'policy': mlp_weights_flat(rng, 6, 4, H=32), 'Q1': mlp_weights_flat(rng, 8, 1, H=32),
            'Q2': mlp_weights_flat(rng, 8, 1, H=32)}
    obs = reset_law_obs(rng, B)
    batch = [obs, rng.uniform(-1, 1, (B, 2)).astype(np.float32), rng.standard_normal(B).astype(np.float32),
             reset_law_obs(rng, B)]
    eps = rng.standard_normal((25, B)).astype(np.float32)
    return O.Cfg(H=32), flat, batch, eps, np.array([0.3, 0.7])


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from mpg_amd import dist as D
    r, w_, _ = D.init_from_env(backend='gloo')
    assert (r, w_) == (rank, world) and D.world_size() == world
    cfg, flat, batch, eps, w = _inputs()
    B = batch[0].shape[0]
    lo, hi = rank * B // world, (rank + 1) * B // world          # contiguous shard of the start states / replay rows
    shard = [b[lo:hi] for b in batch]
    buf = _partials(cfg, flat, shard, eps[:, lo:hi], w, 1.0 / B)
    D.all_reduce_sum_(buf)                                        # the ONE exchange step
    D.barrier()
    t = D.max_over_ranks(float(rank + 1))
    if rank == 0:
        q.put((buf.numpy(), t))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_partials_allreduce_equals_full_batch_gradient():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, tmax = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert tmax == float(world)                                   # max-over-ranks timing helper
    cfg, flat, batch, eps, w = _inputs()
    B = batch[0].shape[0]
    ref = _partials(cfg, flat, batch, eps, w, 1.0 / B).numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-12)
    # clip AFTER the reduce (non-linear): per-network tf.clip_by_global_norm on the reduced buffer == single process
    nq = 8 * 32 + 32 + 32 * 32 + 32 + 32 + 1
    for sl in (slice(0, nq), slice(nq, 2 * nq), slice(2 * nq, got.size - 4)):
        g1, n1 = O.clip_by_global_norm([torch.as_tensor(got[sl])], 3.0)
        g2, n2 = O.clip_by_global_norm([torch.as_tensor(ref[sl])], 3.0)
        np.testing.assert_allclose(g1[0].numpy(), g2[0].numpy(), rtol=1e-10, atol=1e-12)


def test_single_process_dist_helpers_are_noops():
    from mpg_amd import dist as D
    t = torch.arange(4, dtype=torch.float32)
    assert D.world_size() == 1 and torch.equal(D.all_reduce_sum_(t.clone()), t) and D.max_over_ranks(3.5) == 3.5
    D.barrier()


def test_twoshot_slices_partition_the_buffer():
    """mpg_amd.dist.twoshot_slices: the reduce-scatter / all-gather slices of the two-shot exchange are contiguous, disjoint,
    cover the buffer exactly and are 64-float aligned - for the gradient buffers of all four learners, ragged lengths and every world
    size up to 8 (trailing slices may be empty: more ranks than 64-float chunks)."""
    from mpg_amd.dist import twoshot_slices
    for n in (205318 + 16, 136965 + 16, 1, 63, 64, 65, 127, 4097, 1 << 20):
        for world in range(1, 9):
            sl = twoshot_slices(n, world)
            assert len(sl) == world and sl[0][0] == 0 and sl[-1][1] == n
            for r, (lo, hi) in enumerate(sl):
                assert 0 <= lo <= hi <= n and lo % 64 == 0 or lo == n
                if r:
                    assert lo == sl[r - 1][1]
            assert sum(hi - lo for lo, hi in sl) == n
