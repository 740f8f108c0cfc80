"""Golden vectors of the AMP-observation builder: the reference's OWN build_amp_observations (env/tasks/humanoid_amp.py:280)
imported unmodified from /root/reference/ase and run on seeded inputs (incl. the edge cases of its branches).  Its
quaternion primitives come from the isaacgym restatement in oracle/rl_games_shim (Isaac Gym is not vendored).
    python oracle/make_golden_amp_obs.py  ->  tests/golden/amp_obs.pt"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'rl_games_shim'))
sys.path.insert(0, '/root/reference/ase')

from env.tasks import humanoid_amp as H        # noqa: E402  (reference code)

# sword & shield humanoid (ase/data/assets/mjcf/amp_humanoid_sword_shield.xml via env/tasks/humanoid.py:_setup_character_props)
DOF_OFFSETS = [0, 3, 6, 9, 10, 13, 16, 17, 20, 21, 24, 27, 28, 31]


def main():
    g = torch.Generator().manual_seed(4242)
    N, D, K = 96, 31, 6
    q = torch.randn(N, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True)
    q[0] = torch.tensor([0.0, 0.0, 0.0, 1.0])                    # identity
    q[1] = torch.tensor([0.0, 0.0, 1.0, 0.0])                    # half turn about z (heading = pi)
    q[2] = torch.tensor([0.0, 0.7071067811865476, 0.0, 0.7071067811865476])   # x axis points down: heading from (0, 0)
    root_pos = torch.randn(N, 3, generator=g) * torch.tensor([3.0, 3.0, 0.3]) + torch.tensor([0.0, 0.0, 0.9])
    dof_pos = torch.randn(N, D, generator=g) * 0.8
    dof_pos[3] = 0.0                                             # zero exponential maps: default axis branch
    dof_pos[4, 0:3] = torch.tensor([1e-6, -2e-6, 1e-6])          # below the 1e-5 threshold
    dof_pos[5, 0:3] = torch.tensor([2.5, 2.5, 2.5])              # |e| > pi: the angle wraps
    dof_pos[6, 9] = 3.5                                          # 1-dof joint beyond pi
    inputs = {'root_pos': root_pos, 'root_rot': q, 'root_vel': torch.randn(N, 3, generator=g) * 2,
              'root_ang_vel': torch.randn(N, 3, generator=g) * 3, 'dof_pos': dof_pos,
              'dof_vel': torch.randn(N, D, generator=g) * 5, 'key_body_pos': torch.randn(N, K, 3, generator=g) + root_pos.unsqueeze(1)}
    G = {'dof_offsets': DOF_OFFSETS, 'inputs': inputs, 'outputs': {}}
    for local_root in (True, False):
        for root_h in (True, False):
            out = H.build_amp_observations(inputs['root_pos'], inputs['root_rot'], inputs['root_vel'], inputs['root_ang_vel'],
                                           inputs['dof_pos'], inputs['dof_vel'], inputs['key_body_pos'], local_root, root_h,
                                           6 * (len(DOF_OFFSETS) - 1), DOF_OFFSETS)
            G['outputs'][(local_root, root_h)] = out.clone()
    path = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'amp_obs.pt')
    torch.save(G, path)
    print('wrote', path, tuple(out.shape))


if __name__ == '__main__':
    main()
