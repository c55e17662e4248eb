"""Full training-state checkpoint / resume.

The reference saves only network weights (`tf.train.Checkpoint`, policy.py:98-110) and the preprocessor's parameters
(preprocessor.py:176-182, worker.py:66-73): Adam moments, optimizer step counters, the iteration, the replay buffer and
every random stream are lost on restart (train_script.py:819-833 reloads weights + ppc params only).  This module
writes everything the single-process loop needs so that `load -> K steps` is BIT-IDENTICAL to having continued for K
steps (every random draw on the path is Philox(seed, counter), every reduction has a fixed order).

Wire format: one uncompressed `.npz` (zip of `.npy`, readable without this package):
  meta                     JSON (utf-8 bytes): format version, class names, names of the networks, scalars / counters
  policy/{params,targets,m,v,nonfinite}      flat float32 in the reference's Keras order Q1,(Q2),policy
  worker/{env_state,obs,done}                SoA env state [8, num_agent], last observation, done mask
  learner/batch_*                            the minibatch cached across `num_batch_reuse` calls (+ its targets)
  learner/noise_gen                          torch generator state of the host-drawn noise stream (NADP / TD3)
  buffer/{obs,act,rew,obs2,done}             the first `size` ring rows
  buffer/{it_sum,it_min,stamp,max_priority}  prioritized replay only (float64 trees in the reference's heap layout)
"""
import json

import numpy as np
import torch

FORMAT_VERSION = 1


def _np(t):
    return t.detach().cpu().numpy()


def state_of(optimizer):
    """-> (meta dict, {name: numpy array})"""
    from .buffer import PrioritizedReplayBuffer
    w, ln, rb = optimizer.worker, optimizer.learner, optimizer.replay_buffer
    pw = w.policy_with_value
    arrays = {'policy/' + k: _np(getattr(pw, k)) for k in ('params', 'targets', 'm', 'v', 'nonfinite')}
    arrays.update({'worker/env_state': _np(w.env._state), 'worker/obs': _np(w.obs), 'worker/done': _np(w.env.done)})
    for k, v in ln.batch_data.items():
        arrays['learner/' + k] = _np(v)
    arrays['learner/noise_gen'] = _np(ln._noise_gen.get_state())
    n = rb._size
    for k in ('obs', 'act', 'rew', 'obs2', 'done'):
        arrays['buffer/' + k] = _np(getattr(rb, k)[:n])
    per = isinstance(rb, PrioritizedReplayBuffer)
    if per:
        arrays.update({'buffer/it_sum': _np(rb._it_sum), 'buffer/it_min': _np(rb._it_min), 'buffer/stamp': _np(rb._stamp),
                       'buffer/max_priority': _np(rb._max_priority)})
    meta = dict(
        format_version=FORMAT_VERSION, learner_cls=type(ln).__name__, buffer_cls=type(rb).__name__, names=list(pw.names),
        learner_version=getattr(ln.args, 'learner_version', None),
        optimizer=dict(iteration=optimizer.iteration, num_sampled_steps=optimizer.num_sampled_steps),
        policy=dict(opt_steps={k: int(v) for k, v in pw.opt_steps.items()}),
        worker=dict(seed=w.seed, noise_ctr=w._noise_ctr, env_seed=w.env.seed, env_ctr=w.env._ctr, num_sample=w.num_sample,
                    sample_times=w.sample_times, iteration=w.iteration, env_initialised=bool(w.env._initialised)),
        learner=dict(seed=ln.seed, counter=ln.counter),
        buffer=dict(seed=rb.seed, next_idx=rb._next_idx, size=rb._size, maxsize=rb._maxsize, replay_times=rb.replay_times,
                    prioritized=per))
    return meta, arrays


def save_checkpoint(path, optimizer):
    """Write the whole training state of a SingleProcessOffPolicyOptimizer to `path` (.npz)."""
    meta, arrays = state_of(optimizer)
    torch.cuda.synchronize()
    with open(path, 'wb') as f:
        np.savez(f, meta=np.frombuffer(json.dumps(meta).encode('utf-8'), dtype=np.uint8), **arrays)
    return path


def load_checkpoint(path, optimizer):
    """Restore a state written by save_checkpoint into an optimizer built with the same arguments.  Tensors are
    overwritten in place, so the native step driver's context (raw device pointers) stays valid."""
    from .buffer import PrioritizedReplayBuffer
    z = np.load(path)
    meta = json.loads(bytes(z['meta']).decode('utf-8'))
    if meta['format_version'] != FORMAT_VERSION:
        raise ValueError('checkpoint format %r, this build reads %d' % (meta['format_version'], FORMAT_VERSION))
    w, ln, rb = optimizer.worker, optimizer.learner, optimizer.replay_buffer
    pw = w.policy_with_value
    if meta['names'] != list(pw.names) or meta['learner_cls'] != type(ln).__name__ or meta['buffer_cls'] != type(rb).__name__:
        raise ValueError('checkpoint was written by a different configuration: %s / %s / %s' %
                         (meta['learner_cls'], meta['buffer_cls'], meta['names']))
    if meta['buffer']['maxsize'] != rb._maxsize or z['worker/env_state'].shape != tuple(w.env._state.shape):
        raise ValueError('checkpoint buffer capacity / agent count differ from this run')
    dev = pw.device

    def put(dst, name):
        dst.copy_(torch.from_numpy(z[name]).to(dev))

    for k in ('params', 'targets', 'm', 'v', 'nonfinite'):
        put(getattr(pw, k), 'policy/' + k)
    pw.opt_steps = {k: int(v) for k, v in meta['policy']['opt_steps'].items()}
    pw.refresh_weight_cache()
    put(w.env._state, 'worker/env_state')
    m = meta['worker']
    w.seed, w._noise_ctr, w.env.seed, w.env._ctr = m['seed'], m['noise_ctr'], m['env_seed'], m['env_ctr']
    w.num_sample, w.sample_times, w.iteration, w.env._initialised = m['num_sample'], m['sample_times'], m['iteration'], m['env_initialised']
    w.obs = w.env.obs = torch.from_numpy(z['worker/obs']).to(dev)
    w.env.done = torch.from_numpy(z['worker/done']).to(dev)
    ln.seed, ln.counter = meta['learner']['seed'], meta['learner']['counter']
    ln._noise_gen.set_state(torch.from_numpy(z['learner/noise_gen']))
    for name in z.files:
        if name.startswith('learner/batch_'):
            k = name[len('learner/'):]
            if k in ln.batch_data and ln.batch_data[k].shape == z[name].shape:
                put(ln.batch_data[k], name)
            else:
                ln.batch_data[k] = torch.from_numpy(z[name]).to(dev)
    b = meta['buffer']
    n = b['size']
    for k in ('obs', 'act', 'rew', 'obs2', 'done'):
        getattr(rb, k)[:n].copy_(torch.from_numpy(z['buffer/' + k]).to(dev))
    rb.seed, rb._next_idx, rb._size, rb.replay_times = b['seed'], b['next_idx'], n, b['replay_times']
    if isinstance(rb, PrioritizedReplayBuffer):
        for k in ('it_sum', 'it_min', 'stamp', 'max_priority'):
            put(getattr(rb, '_' + k), 'buffer/' + k)
    optimizer.iteration = meta['optimizer']['iteration']
    optimizer.num_sampled_steps = meta['optimizer']['num_sampled_steps']
    if getattr(optimizer, '_fused', None) is not None:
        optimizer._fused.reload()
    return meta
