"""Error behaviour of the C ABI (include/mpg_hip.h: "0 on success, an MPG_ERR_* code otherwise, text behind mpg_last_error()").
Every entry point validates its arguments BEFORE it touches the device, so these refusals can be checked without a GPU: a bad
argument must come back as a non-zero code with a message that names the entry point - never as a crash, and never as a launch.
(The reference's counterparts fail with Python exceptions - an assert in segment_tree.py:40, numpy shape errors in buffer.py -
which is what mpg_amd._lib.call turns these codes into.)"""
import ctypes

import pytest

from mpg_amd import _lib as L

NULL = ctypes.c_void_p(0)
FAKE = ctypes.c_void_p(0x1000)          # a non-null pointer that is never dereferenced: the call must be refused before any launch
I, U64, F, SZ = ctypes.c_int, ctypes.c_uint64, ctypes.c_float, ctypes.c_size_t

CASES = [
    # (entry point, arguments, what is wrong)
    ('mpg_env_step', (I(7), I(16), I(6), FAKE, FAKE, FAKE, FAKE, FAKE, NULL, NULL), 'unknown env kind'),
    ('mpg_env_step', (I(0), I(0), I(6), FAKE, FAKE, FAKE, FAKE, FAKE, NULL, NULL), 'no agents'),
    ('mpg_env_step', (I(0), I(16), I(5), FAKE, FAKE, FAKE, FAKE, FAKE, NULL, NULL), 'obs_dim below the six base entries'),
    ('mpg_env_step', (I(0), I(16), I(6 + 11), FAKE, FAKE, FAKE, FAKE, FAKE, NULL, NULL), 'more look-ahead entries than MPG_ENV_MAX_FUTURE'),
    ('mpg_env_step', (I(0), I(16), I(6), NULL, FAKE, FAKE, FAKE, FAKE, NULL, NULL), 'null state'),
    ('mpg_env_reset', (I(0), I(16), I(6), FAKE, NULL, U64(1), U64(0), NULL, NULL), 'null obs'),
    ('mpg_env_reset_from_obs', (I(0), I(16), I(6), FAKE, NULL, NULL), 'null init_obs'),
    ('mpg_env_step_store_reset', (I(0), I(64), I(6), FAKE, FAKE, I(32), I(0), FAKE, FAKE, FAKE, FAKE, FAKE, U64(1), U64(0), FAKE, NULL, NULL),
     'ring smaller than one batch of transitions'),
    ('mpg_env_step_store_reset', (I(0), I(16), I(6), FAKE, FAKE, I(64), I(64), FAKE, FAKE, FAKE, FAKE, FAKE, U64(1), U64(0), FAKE, NULL, NULL),
     'next_idx outside the ring'),
    ('mpg_sum_slots', (NULL, I(2), I(64), FAKE, NULL), 'null slots'),
    ('mpg_sum_slots', (FAKE, I(0), I(64), FAKE, NULL), 'no slots'),
    ('mpg_sum_slots_strided', (FAKE, I(2), SZ(32), I(64), FAKE, NULL), 'slice longer than the slot stride'),
    ('mpg_replay_add', (I(64), I(0), I(128), I(6), I(2), FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, NULL), 'more rows than the ring holds'),
    ('mpg_replay_add', (I(64), I(-1), I(16), I(6), I(2), FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, NULL), 'negative ring position'),
    ('mpg_replay_gather', (I(16), FAKE, I(17), I(2), FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, NULL, NULL), 'obs_dim beyond 16'),
    ('mpg_replay_gather', (I(16), NULL, I(6), I(2), FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, NULL, NULL), 'null index array'),
    ('mpg_uniform_indices', (I(0), I(16), U64(1), U64(1), FAKE, NULL), 'empty storage (the reference: randint(0, -1) raises)'),
    ('mpg_uniform_indices', (I(100), I(16), U64(1), U64(1), NULL, NULL), 'null output'),
    ('mpg_per_init', (FAKE, FAKE, FAKE, I(1000), NULL), 'capacity not a power of two (segment_tree.py:40 asserts the same)'),
    ('mpg_per_init', (NULL, FAKE, FAKE, I(1024), NULL), 'null tree'),
    ('mpg_normal_fill', (I(0), U64(1), U64(1), FAKE, NULL), 'nothing to fill'),
    ('mpg_mlp_forward', (NULL, I(6), I(4), I(2), I(0), I(16), FAKE, NULL, I(0), FAKE, NULL, NULL), 'null parameters'),
    ('mpg_mlp_forward', (FAKE, I(3), I(4), I(2), I(0), I(16), FAKE, NULL, I(0), FAKE, NULL, NULL), 'a network shape that is not built'),
    ('mpg_mlp_forward', (FAKE, I(6), I(4), I(2), I(0), I(0), FAKE, NULL, I(0), FAKE, NULL, NULL), 'no rows'),
    ('mpg_policy_action', (NULL, FAKE, I(16), FAKE, F(0.0), U64(0), U64(0), FAKE, NULL), 'null configuration'),
    ('mpg_clip_by_global_norm', (FAKE, FAKE, I(0), F(3.0), FAKE, NULL, NULL, NULL), 'no segments'),
    ('mpg_prof_read', (NULL, I(0), NULL, NULL), 'null profiler handle'),
]


SHARED = {'a network shape that is not built': 'unsupported network shape', 'null parameters': 'launch_forward', 'no rows': 'launch_forward'}


@pytest.mark.parametrize('name,args,why', CASES, ids=['%s-%s' % (c[0], c[2].split(' (')[0].replace(' ', '_')) for c in CASES])
def test_bad_argument_is_refused_with_a_code_and_a_message(name, args, why):
    lib = L.lib()
    rc = getattr(lib, name)(*args)
    assert rc != 0, '%s accepted: %s' % (name, why)
    msg = lib.mpg_last_error().decode()
    # the text names the entry point (or, for a refusal inside a shared launcher, what was refused)
    assert msg and (name.replace('mpg_', '') in msg.replace('mpg_', '') or SHARED.get(why, '\0') in msg), (name, msg)
    with pytest.raises(L.MpgError, match=name):          # ... and the binding raises (no silent fallback anywhere)
        L.call(name, *args)


def test_the_message_belongs_to_the_calling_thread():
    """mpg_last_error() is thread-local (the one piece of state the library keeps): a refusal in another thread
    does not replace this thread's text"""
    import threading
    lib = L.lib()
    assert lib.mpg_sum_slots(NULL, I(2), I(64), FAKE, NULL) != 0
    mine = lib.mpg_last_error().decode()
    seen = {}

    def other():
        assert lib.mpg_uniform_indices(I(0), I(16), U64(1), U64(1), FAKE, NULL) != 0
        seen['msg'] = lib.mpg_last_error().decode()
    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert 'uniform_indices' in seen['msg'] and 'sum_slots' in mine
    assert lib.mpg_last_error().decode() == mine
