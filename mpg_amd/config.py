"""Default hyper-parameters = the reference's argparse defaults (train_scripts/train_script.py:177-306 MPG,
:551-670 TD3; train_scripts/train_script4mujoco.py:296-411 NADP on InvertedPendulumConti-v0), under the same names,
so `Namespace` objects are interchangeable with the reference's `args`."""
import argparse


def default_args(alg='MPG-v2', env_id=None, **overrides):
    pend = alg == 'NADP' if env_id is None else env_id == 'InvertedPendulumConti-v0'
    env_id = env_id or ('InvertedPendulumConti-v0' if pend else 'PathTracking-v0')
    d = dict(
        policy_type='PolicyWithQs', worker_type='OffPolicyWorker', buffer_type='normal', optimizer_type='SingleProcessOffPolicy',
        env_id=env_id, num_agent=8 if not pend else 1, num_future_data=0,
        alg_name=alg.split('-')[0], learner_version=alg, sample_num_in_learner=25, M=1, deriv_interval_policy=False,
        num_rollout_list_for_policy_update=[0, 25] if alg.startswith('MPG') else [25],
        num_rollout_list_for_q_estimation=[] if alg.startswith('MPG') else [25],
        eta=0.1, rule_based_bias_total_ite=9000, gamma=0.98, gradient_clip_norm=3.,
        num_batch_reuse=10 if alg == 'MPG-v1' else 1,
        batch_size=512, explore_sigma=None if alg == 'NADP' else 0.1,
        max_buffer_size=500000, replay_starts=3000, replay_batch_size=256, replay_alpha=0.6, replay_beta=0.4,
        obs_dim=4 if pend else 6, act_dim=1 if pend else 2,
        value_model_cls='MLP', value_num_hidden_layers=2, value_num_hidden_units=256, value_hidden_activation='elu',
        value_lr_schedule=[8e-5, 100000, 8e-6],
        policy_model_cls='MLP', policy_num_hidden_layers=2, policy_num_hidden_units=256, policy_hidden_activation='elu',
        policy_out_activation='linear' if pend else 'tanh', policy_lr_schedule=[3e-5, 100000, 3e-6],
        alpha=None, alpha_lr_schedule=None, policy_only=False, double_Q=alg in ('MPG-v2', 'TD3'), target=True, tau=0.005,
        delay_update=1 if alg == 'NADP' else 2, deterministic_policy=True, action_range=3. if pend else None,
        obs_ptype='scale', obs_scale=[0.001, 1 / 3, 0.1, 0.5] if pend else [1., 1., 2., 1., 2.4, 1 / 1200],
        rew_ptype='scale', rew_scale=1. if pend else 0.01, rew_shift=0.,
        policy_smoothing_sigma=0.2, policy_smoothing_clip=0.5,
        max_iter=100000, seed=0, init_seed=0)
    d.update(overrides)
    if not pend and d['num_future_data'] and 'obs_dim' not in overrides:       # train_script.py:146-147, 794-811
        d['obs_dim'] = 6 + d['num_future_data']
        if 'obs_scale' not in overrides:
            d['obs_scale'] = d['obs_scale'] + [1.] * d['num_future_data']
    return argparse.Namespace(**d)
