import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpg_amd.buffer import ReplayBuffer
from mpg_amd.config import default_args
from mpg_amd.evaluator import Evaluator
from mpg_amd.learners import MPGLearner
from mpg_amd.optimizer import SingleProcessOffPolicyOptimizer
from mpg_amd.policy import PolicyWithQs
from mpg_amd.worker import OffPolicyWorker

def run(tag, iters, every, **kw):
    args = default_args('MPG-v2', num_eval_agent=256, **kw)
    worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, 0)
    learner = MPGLearner(PolicyWithQs, args)
    rb = ReplayBuffer(args, 0)
    opt = SingleProcessOffPolicyOptimizer(worker, learner, rb, None, args, sampling_interval=SI)
    ev = Evaluator(PolicyWithQs, args.env_id, args)
    ev.share_policy(worker.policy_with_value)
    t0 = time.perf_counter()
    for it in range(0, iters + 1, every):
        m = ev.run_evaluation(it)
        torch.cuda.synchronize()
        print(json.dumps(dict(run=tag, iteration=it, wall_s=round(time.perf_counter() - t0, 1), episode_return=round(m['episode_return'], 2),
                              delta_y_rms=round(m['delta_y_mse'], 3), delta_phi_rms=round(m['delta_phi_mse'], 4),
                              nonfinite=int(worker.policy_with_value.nonfinite.sum().item()))), flush=True)
        if it < iters:
            for _ in range(every):
                opt.step()
SI = 10
run('reference defaults: 8 agents, batch 256, sample every 10th iteration, 100k iterations', 100000, 10000)
SI = 1
run('bench config: 4096 agents, batch 4096, sample every iteration, 20k iterations', 20000, 4000, num_agent=4096, batch_size=4096,
    replay_batch_size=4096, replay_starts=16384)
