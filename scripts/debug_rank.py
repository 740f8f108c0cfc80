import faulthandler, os, sys, torch
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
rank = int(sys.argv[1]) if len(sys.argv) > 1 else 1
agent, cfg, spec = bench.make_agent('cuda:0', 'bf16', False, 2, rank)
print('agent built', flush=True)
eng = agent.engine
eng._allreduce_stats = lambda: None
eng._allreduce_grads = lambda: None
with torch.no_grad():
    exp = agent.vec_env.experience(agent._cpu_policy())
    for k, v in exp.items():
        if k in agent.experience:
            agent.experience[k].copy_(v.to('cuda:0'))
    agent._init_amp_demo_buf()
torch.cuda.synchronize(); print('rollout done', flush=True)
batch = agent._play_steps_tail(); torch.cuda.synchronize(); print('tail done', flush=True)
info = agent.update(batch, max_steps=2); torch.cuda.synchronize(); print('update done', {k: float(v[-1]) for k, v in info.items() if torch.is_tensor(v[-1]) and v[-1].numel() == 1})
