"""The numerical envelope of the split-fp16 hidden-layer engine (include/mpg_hip.h, "Numerical envelope"; csrc/mlp_core.h):
large-magnitude operands stay within the float32 tolerance up to the documented limits, and beyond them the library
reports a defined error through the caller's status word - it never hands an fp16 infinity on silently.

What the reference does with such inputs: TensorFlow computes in float32 throughout (model.py:39-43), there is no range
to leave.  So inside the envelope the comparison is the usual one (float64 oracle), and outside it the contract is the
status bit + finite outputs."""
import numpy as np
import pytest
import torch

from oracle import mpg_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).to(DEV)


def net_parts(rng, din, dout, w1=0.4, w2=1.4 / 16, w3=0.1):
    return [rng.standard_normal((din, 256)) * w1, rng.standard_normal(256) * 0.1, rng.standard_normal((256, 256)) * w2,
            rng.standard_normal(256) * 0.1, rng.standard_normal((256, dout)) * w3, rng.standard_normal(dout) * 0.1]


def flat(parts):
    return np.concatenate([np.asarray(p).ravel() for p in parts]).astype(np.float32)


def status_cfg():
    """a cfg with its own status word (what PolicyWithQs does)"""
    from mpg_amd import ops
    cfg = ops.make_cfg()
    st = torch.zeros(1, dtype=torch.int32, device=DEV)
    cfg.status = st.data_ptr()
    return cfg, st


def q_oracle(flat_w, obs, act, dtype):
    ocfg = O.Cfg()
    nets = O.Nets(ocfg, {'Q1': flat_w}, dtype=dtype)
    po = O.process_obses(ocfg, torch.as_tensor(obs).to(dtype))
    return nets, po, torch.as_tensor(act).to(dtype)


@pytest.mark.parametrize('factor', [8.0, 64.0])
def test_hidden_kernel_scaled_up_stays_within_tolerance(factor):
    """W2 x 8 and x 64 (entries up to ~25, far above any initialisation, inside |W| < 1023.5): forward values and the critic's
    loss gradient against the float64 oracle under the usual bars; the status word stays clear."""
    from mpg_amd import ops
    rng = np.random.Generator(np.random.PCG64(int(factor)))
    parts = net_parts(rng, 8, 1, w2=factor * 1.4 / 16)
    w = flat(parts)
    B = 512
    obs = (rng.standard_normal((B, 6)) * np.array([3, 1, .5, 1, .5, 300])).astype(np.float32)
    act = rng.uniform(-1, 1, (B, 2)).astype(np.float32)
    cfg, st = status_cfg()
    x = torch.cat([dev(obs), dev(act)], 1).contiguous()
    sc = [cfg.obs_scale[i] for i in range(6)]
    q = ops.mlp_forward(dev(w), 8, 1, 1, 0, x, in_scale=sc, n_scaled=6).cpu().numpy()[:, 0]
    nets, po, a64 = q_oracle(w, obs, act, torch.float64)
    qref = nets.q('Q1', po, a64).detach().numpy()
    assert np.abs(q - qref).max() <= 2e-5 * np.abs(qref).max()
    y = (qref + rng.standard_normal(B)).astype(np.float32)
    _, g, _ = ops.q_loss_grad(cfg, dev(w), dev(obs), dev(act), dev(y))
    loss = 0.5 * torch.mean((nets.q('Q1', po, a64) - torch.as_tensor(y).double()) ** 2)
    gref = np.concatenate([t.numpy().ravel() for t in torch.autograd.grad(loss, nets.w['Q1'])])
    nets32, po32, a32 = q_oracle(w, obs, act, torch.float32)
    loss32 = 0.5 * torch.mean((nets32.q('Q1', po32, a32) - torch.as_tensor(y)) ** 2)
    g32 = np.concatenate([t.numpy().ravel() for t in torch.autograd.grad(loss32, nets32.w['Q1'])])
    got = g.cpu().numpy()
    o = 0
    for shp in O.mlp_shapes(8, 256, 1):
        n = int(np.prod(shp))
        e_got, e_ref = rel_l2(got[o:o + n], gref[o:o + n]), rel_l2(g32[o:o + n], gref[o:o + n])
        if n >= 8:       # db3 is ONE number, the sum of B signed errors: ill-conditioned for every float32 implementation
            assert rel_l2(got[o:o + n], g32[o:o + n]) <= 1e-4, (shp, 'vs float32 oracle')
            assert e_got <= 4 * e_ref + 1e-6, (shp, 'vs float64: got %.2e, float32 oracle %.2e' % (e_got, e_ref))
        o += n
    assert int(st.item()) == 0


def _first_layer_push(target_h):
    """a critic whose first-layer activations reach ~target_h on a few units: one input column of W1 is scaled up"""
    rng = np.random.Generator(np.random.PCG64(7))
    parts = net_parts(rng, 8, 1)
    B = 256
    obs = (rng.standard_normal((B, 6)) * np.array([3, 1, .5, 1, .5, 300])).astype(np.float32)
    act = rng.uniform(-1, 1, (B, 2)).astype(np.float32)
    x = np.concatenate([obs * np.array([1., 1., 2., 1., 2.4, 1 / 1200], np.float32), act], 1).astype(np.float64)
    z = x @ parts[0] + parts[1]
    parts[0] = parts[0] * (target_h / z.max())
    parts[1] = parts[1] * (target_h / z.max())
    return flat(parts), obs, act, float((x @ parts[0] + parts[1]).max())


def test_activations_up_to_the_limit_are_exact_and_beyond_it_reported():
    """|h1| < 4094 is the envelope (x * 16 in fp16).  Just inside it (max h1 ~ 3900) the forward values hold the usual bar and
    nothing is reported; beyond it (max h1 ~ 6000) MPG_STATUS_ACTIVATION_RANGE is set by the same call - the affected rows are
    garbage by contract, the others still exact - and PolicyWithQs.check_status() turns the bit into an exception."""
    from mpg_amd import ops
    from mpg_amd._lib import MpgError
    from mpg_amd.policy import PolicyWithQs
    for target, expect in ((3900.0, 0), (6000.0, ops.STATUS_ACTIVATION_RANGE)):
        w, obs, act, hmax = _first_layer_push(target)
        cfg, st = status_cfg()
        y = np.zeros(obs.shape[0], np.float32)
        _, g, td = ops.q_loss_grad(cfg, dev(w), dev(obs), dev(act), dev(y), want_td=True)
        torch.cuda.synchronize()
        assert int(st.item()) == expect, (target, hmax, int(st.item()))
        nets, po, a64 = q_oracle(w, obs, act, torch.float64)
        qref = nets.q('Q1', po, a64).detach().numpy()
        x64 = torch.cat([po, a64], 1)
        h1 = torch.nn.functional.elu(x64 @ nets.w['Q1'][0] + nets.w['Q1'][1]).detach().numpy()
        ok_rows = h1.max(axis=1) < 4094.0
        assert ok_rows.sum() > 0
        q = td.cpu().numpy()                       # td = Q - y with y = 0
        assert np.all(np.isfinite(q[ok_rows]))
        assert np.abs(q[ok_rows] - qref[ok_rows]).max() <= 2e-5 * np.abs(qref).max(), target
        if expect == 0:
            assert ok_rows.all()
    # the python owner of a status word raises on it and clears it
    pw = PolicyWithQs(6, 2, device=DEV)
    assert pw.check_status() == 0
    w, obs, act, _ = _first_layer_push(6000.0)
    pw.net('Q1').copy_(dev(w))
    pw.refresh_weight_cache()
    pw.compute_Q1(dev(obs), dev(act))            # mpg_mlp_forward has no cfg: the cfg-carrying entry points report
    ops.q_loss_grad(pw.cfg, pw.net('Q1'), dev(obs), dev(act), dev(np.zeros(obs.shape[0], np.float32)))
    with pytest.raises(MpgError, match='numerical envelope'):
        pw.check_status()
    assert pw.check_status() == 0                 # cleared by the read


def test_parameter_beyond_the_limit_is_clamped_in_the_image_and_reported():
    """|W| < 1023.5 is the envelope of the packed images (W * 64 in fp16).  An entry of 3000 is packed as 65504 / 64 (finite)
    and MPG_STATUS_PARAMETER_RANGE is set - by mpg_weight_cache_pack and by the Adam update that writes such a value."""
    from mpg_amd import ops
    from mpg_amd.policy import PolicyWithQs
    pw = PolicyWithQs(6, 2, device=DEV)
    torch.cuda.synchronize()
    assert int(pw.status.item()) == 0
    q1 = pw.net('Q1')
    w2_off = 8 * 256 + 256
    q1[w2_off + 5 * 256 + 9] = 3000.0
    pw.refresh_weight_cache()
    torch.cuda.synchronize()
    assert int(pw.status.item()) & ops.STATUS_PARAMETER_RANGE
    rng = np.random.Generator(np.random.PCG64(3))
    obs, act = dev(rng.standard_normal((64, 6))), dev(rng.uniform(-1, 1, (64, 2)))
    q = pw.compute_Q1(obs, act)
    assert torch.isfinite(q).all()
    # the same through the optimizer: a huge gradient on one hidden-kernel entry drives it past the limit
    pw.status.zero_()
    q1[w2_off + 5 * 256 + 9] = 1023.0
    pw.refresh_weight_cache()
    torch.cuda.synchronize()
    assert int(pw.status.item()) == 0
    pw.schedules = {n: (10.0, 100000, 10.0) for n in pw.names}          # lr 10: one Adam step moves the entry by ~10
    grads = torch.zeros(int(pw.offsets[-1]), device=DEV)
    grads[w2_off + 5 * 256 + 9] = -1.0
    pw.apply_gradients(0, grads)
    torch.cuda.synchronize()
    assert float(q1[w2_off + 5 * 256 + 9]) > 1023.5
    assert int(pw.status.item()) & ops.STATUS_PARAMETER_RANGE
    assert torch.isfinite(pw.compute_Q1(obs, act)).all()


@pytest.mark.parametrize('scale', [1e-4, 1.0, 1e4])
def test_weight_gradient_is_scale_invariant(scale):
    """The weight-gradient product scales its dz operand per chunk from the data (max |dL/dz3| of the chunk's rows): TD errors
    of 1e-4 and of 1e+4 give the same relative accuracy against the float64 oracle as O(1) ones (a fixed scale of ~B left the
    small ones with a few bits in their fp16 lo halves: ADVICE r2)."""
    from mpg_amd import ops
    rng = np.random.Generator(np.random.PCG64(11))
    w = flat(net_parts(rng, 8, 1))
    B = 1024
    obs = (rng.standard_normal((B, 6)) * np.array([3, 1, .5, 1, .5, 300])).astype(np.float32)
    act = rng.uniform(-1, 1, (B, 2)).astype(np.float32)
    nets, po, a64 = q_oracle(w, obs, act, torch.float64)
    qref = nets.q('Q1', po, a64).detach().numpy()
    y = (qref + scale * rng.standard_normal(B)).astype(np.float32)
    cfg, st = status_cfg()
    _, g, _ = ops.q_loss_grad(cfg, dev(w), dev(obs), dev(act), dev(y))
    loss = 0.5 * torch.mean((nets.q('Q1', po, a64) - torch.as_tensor(y).double()) ** 2)
    gref = np.concatenate([t.numpy().ravel() for t in torch.autograd.grad(loss, nets.w['Q1'])])
    nets32, po32, a32 = q_oracle(w, obs, act, torch.float32)
    loss32 = 0.5 * torch.mean((nets32.q('Q1', po32, a32) - torch.as_tensor(y)) ** 2)
    g32 = np.concatenate([t.numpy().ravel() for t in torch.autograd.grad(loss32, nets32.w['Q1'])])
    got = g.cpu().numpy()
    o = 0
    for shp in O.mlp_shapes(8, 256, 1):
        n = int(np.prod(shp))
        e_got, e_ref = rel_l2(got[o:o + n], gref[o:o + n]), rel_l2(g32[o:o + n], gref[o:o + n])
        if n >= 8:       # db3 is ONE number, the sum of B signed errors: ill-conditioned for every float32 implementation
            assert e_got <= 4 * e_ref + 1e-6, (scale, shp, 'vs float64: got %.2e, float32 oracle %.2e' % (e_got, e_ref))
        o += n
    assert int(st.item()) == 0


def _stack(alg='MPG-v2', fused=True, **kw):
    from mpg_amd.buffer import ReplayBuffer
    from mpg_amd.config import default_args
    from mpg_amd.learners import MPGLearner
    from mpg_amd.optimizer import SingleProcessOffPolicyOptimizer
    from mpg_amd.policy import PolicyWithQs
    from mpg_amd.worker import OffPolicyWorker
    a = dict(num_agent=64, batch_size=64, replay_batch_size=256, replay_starts=512, max_buffer_size=512)
    a.update(kw)
    args = default_args(alg, **a)
    worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, 0)
    learner = MPGLearner(PolicyWithQs, args)
    rb = ReplayBuffer(args, 0)
    opt = SingleProcessOffPolicyOptimizer(worker, learner, rb, None, args, sampling_interval=10 ** 9, fused=fused)
    assert (opt._fused is not None) == fused
    return opt, worker, learner, rb


@pytest.mark.parametrize('field', ['obs', 'act', 'rew', 'obs2'])
@pytest.mark.parametrize('fused', [True, False])
def test_nan_in_a_replay_row_is_reported_and_zeroes_the_step(field, fused):
    """optimizer.py:357-361: a NaN anywhere in the gradient list makes the reference apply ZERO gradients; in TensorFlow a NaN in
    any field of a replay row reaches that list.  Here the ELU's v_med3_f32 drops a NaN inside the networks, so the learner
    kernels look at their inputs themselves (csrc/fused_kernels.hip row_poison / report_nan): the row's TD error becomes NaN,
    the clip kernel sees a non-finite norm and Adam gets zeros (parameters bit-identical after the step on a fresh optimizer
    state), and MPG_STATUS_NAN is raised for the host to read."""
    from mpg_amd._lib import MpgError
    opt, worker, learner, rb = _stack(fused=fused, nan_check_interval=10 ** 9)
    pw = worker.policy_with_value
    # a ring no larger than two minibatches, every slot poisoned in ONE field of ONE coordinate: whichever rows are drawn, they carry it
    t = dict(obs=rb.obs, act=rb.act, rew=rb.rew, obs2=rb.obs2)[field]
    if t.dim() == 1:
        t[7::16] = float('nan')
    else:
        t[7::16, t.shape[1] - 1] = float('nan')
    before = pw.params.clone()
    tbefore = pw.targets.clone()
    opt.step()
    torch.cuda.synchronize()
    assert int(pw.nonfinite.sum().item()) > 0, 'the clip kernel did not see a non-finite gradient'
    assert torch.equal(pw.params, before), 'a step computed from a NaN row moved the parameters'
    # (Polyak still runs: (1 - tau) t + tau w with t == w re-rounds the last bit)
    assert torch.isfinite(pw.targets).all() and (pw.targets - tbefore).abs().max().item() <= 1e-6
    with pytest.raises(MpgError, match='judge_is_nan'):
        pw.check_status()
    assert pw.check_status() == 0                                  # read and cleared


@pytest.mark.parametrize('fused', [True, False])
def test_optimizer_reads_the_status_word_every_nan_check_interval_iterations(fused):
    """the native step driver never goes through worker.sample(), which was the only place that read the status word; a
    parameter pushed out of the engine's envelope must stop the run within `nan_check_interval` iterations in BOTH branches
    of SingleProcessOffPolicyOptimizer.step (the reference's judge_is_nan stop, worker.py:95-107 / optimizer.py:357-361)."""
    from mpg_amd._lib import MpgError
    opt, worker, learner, rb = _stack(fused=fused, nan_check_interval=4)
    pw = worker.policy_with_value
    for _ in range(4):
        opt.step()                                                 # clean: iterations 1..4 include one check
    w2 = pw.net('Q1')[8 * 256 + 256: 8 * 256 + 256 + 256 * 256]
    w2[300] = 3000.0                                               # beyond |w| < 1023.5
    pw.refresh_weight_cache()                                      # (re)pack reports it ...
    with pytest.raises(MpgError, match='envelope'):
        for _ in range(4):
            opt.step()                                             # ... and the optimizer reads the word within 4 iterations
