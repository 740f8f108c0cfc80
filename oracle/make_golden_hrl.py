"""Golden vectors of the HRL rollout step: the REFERENCE'S OWN ``HRLAgent.env_step`` / ``_compute_llc_action`` /
``_calc_disc_reward`` (learning/hrl_agent.py:45-82,231-249, imported unmodified through oracle/ref_runner.py) over a frozen
reference ``ASEAgent`` as low-level controller, on the seeded synthetic environment.  TEST INFRASTRUCTURE ONLY.

    python oracle/make_golden_hrl.py        # writes tests/golden/hrl_step.pt

The agents are built like the other goldens (object.__new__ + attribute injection); every statement that runs inside
env_step is the reference's.  rl_games' ``preprocess_actions`` / ``rescale_actions`` / ``obs_to_tensors`` are restated in the
shim (parity unpinned at that boundary, see oracle/rl_games_shim/README.md)."""
import copy
import importlib
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from ase_amd.synthetic import EnvSpec, SyntheticVecEnv  # noqa: E402
from oracle import ref_runner  # noqa: E402
from oracle.make_golden import _case  # noqa: E402

OUT = os.environ.get("ASE_GOLDEN_OUT", os.path.join(os.path.dirname(HERE), "tests", "golden"))
TASK, LLC_STEPS, N = 5, 3, 16


def main():
    torch.set_num_threads(4)
    ref = ref_runner.load_reference()
    hrl = importlib.import_module('learning.hrl_agent')
    net_l, cfg_l = _case('ase')
    net_h, cfg_h = _case('ppo')
    obs_l, act_l, amp = 37, 7, 44
    z = cfg_l['latent_dim']
    torch.manual_seed(0)
    LLC = ref_runner.build_ref_agent('ase', net_l, cfg_l, num_envs=N, obs_size=obs_l, act_size=act_l, amp_obs_size=amp, seed=0)
    g = torch.Generator().manual_seed(5)
    # non-trivial running statistics (train-mode passes), then eval
    LLC.running_mean_std.train()
    LLC._amp_input_mean_std.train()
    for _ in range(3):
        LLC.running_mean_std(torch.randn(200, obs_l, generator=g) * 1.5 + 0.3)
        LLC._amp_input_mean_std(torch.randn(200, amp, generator=g) * 0.7 - 0.2)
    LLC.set_eval()
    LLC._amp_input_mean_std.eval()
    LLC.is_tensor_obses = True
    cfg_h = dict(cfg_h)
    cfg_h.update(task_reward_w=0.5, disc_reward_w=0.5)
    base = ref_runner.build_ref_agent('ppo', net_h, cfg_h, num_envs=N, obs_size=obs_l + TASK, act_size=z, seed=1)
    H = object.__new__(hrl.HRLAgent)
    H.__dict__.update(base.__dict__)
    H._latent_dim, H._task_size, H._llc_steps, H._llc_agent = z, TASK, LLC_STEPS, LLC
    H._task_reward_w, H._disc_reward_w = 0.5, 0.5
    H.is_tensor_obses = True
    spec = EnvSpec(num_envs=N, horizon=cfg_h['horizon_length'], obs_size=obs_l, act_size=act_l, amp_obs_size=amp, latent_dim=z,
                   episode_length=4)
    env = SyntheticVecEnv(spec, seed=11, task_obs_size=TASK)
    H.vec_env = env
    H.obs = H.obs_to_tensors(env.reset())
    obs0 = H.obs['obs'].clone()
    actions = torch.randn(N, z, generator=g) * 1.5                       # beyond [-1, 1]: preprocess_actions clamps
    with torch.no_grad():
        llc_action0 = H._compute_llc_action(obs0, H.preprocess_actions(actions)).clone()
        obs1, rewards, dones, infos = H.env_step(actions)                 # REFERENCE CODE: learning/hrl_agent.py:45-82
    G = {'llc': {'net': net_l, 'cfg': cfg_l, 'sd': {k: v.detach().clone() for k, v in LLC.model.state_dict().items()},
                 'running_mean_std': copy.deepcopy(LLC.running_mean_std.state_dict()),
                 'amp_input_mean_std': copy.deepcopy(LLC._amp_input_mean_std.state_dict()),
                 'reward_mean_std': copy.deepcopy(LLC.value_mean_std.state_dict())},
         'hlc': {'net': net_h, 'cfg': cfg_h}, 'spec': dict(num_envs=N, obs_size=obs_l, act_size=act_l, amp_obs_size=amp, task=TASK),
         'env_seed': 11, 'episode_length': 4, 'llc_steps': LLC_STEPS, 'obs0': obs0, 'actions': actions,
         'llc_action0': llc_action0, 'env_last_actions': env.last_actions.clone(), 'obs1': obs1['obs'].clone(),
         'rewards': rewards.clone(), 'dones': dones.clone(), 'terminate': infos['terminate'].clone(),
         'disc_rewards': infos['disc_rewards'].clone()}
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, 'hrl_step.pt')
    torch.save(G, path)
    print('wrote', path, '%.2f MB' % (os.path.getsize(path) / 1e6), 'dones', int(dones.sum()), 'terminate', int(infos['terminate'].sum()))


if __name__ == '__main__':
    main()
