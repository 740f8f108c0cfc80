"""Golden vectors of the motion-clip sampler: the reference's OWN MotionLib (utils/motion_lib.py, poselib loader) on two
shipped clips, get_motion_state at seeded (motion id, time) pairs incl. the clip ends.  Quaternion primitives: the
isaacgym restatement of oracle/rl_games_shim.     python oracle/make_golden_motion.py -> tests/golden/motion_state.pt"""
import os
import sys
import tempfile

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'rl_games_shim'))
sys.path.insert(0, '/root/reference/ase')

from utils.motion_lib import MotionLib        # noqa: E402  (reference code)

DOF_BODY_IDS = [1, 2, 3, 4, 5, 7, 8, 11, 12, 13, 14, 15, 16]            # env/tasks/humanoid.py:191-192
DOF_OFFSETS = [0, 3, 6, 9, 10, 13, 16, 17, 20, 21, 24, 27, 28, 31]
KEY_BODY_IDS = [5, 10, 13, 16, 6, 9]
CLIPS = ['RL_Avatar_Atk_2xCombo01_Motion.npy', 'RL_Avatar_Atk_2xCombo02_Motion.npy']


def main():
    d = '/root/reference/ase/data/motions/reallusion_sword_shield'
    with tempfile.TemporaryDirectory() as tmp:
        y = os.path.join(tmp, 'two.yaml')
        with open(y, 'w') as f:
            f.write('motions:\n' + ''.join(f'  - file: "{os.path.join(d, c)}"\n    weight: 0.5\n' for c in CLIPS))
        ml = MotionLib(motion_file=y, dof_body_ids=DOF_BODY_IDS, dof_offsets=DOF_OFFSETS, key_body_ids=KEY_BODY_IDS, device='cpu')
    g = torch.Generator().manual_seed(77)
    n = 200
    ids = torch.randint(0, ml.num_motions(), (n,), generator=g)
    lens = ml._motion_lengths[ids]
    t = torch.rand(n, generator=g) * lens
    t[0], t[1], t[2] = 0.0, lens[1], lens[2] + 0.5                       # start, exact end, past the end (clipped phase)
    t[3] = ml._motion_dt[ids[3]] * 7                                     # exactly on a frame
    out = ml.get_motion_state(ids, t)
    G = {'clips': {'gts': ml.gts, 'grs': ml.grs, 'lrs': ml.lrs, 'grvs': ml.grvs, 'gravs': ml.gravs, 'dvs': ml.dvs,
                   'lengths': ml._motion_lengths, 'num_frames': ml._motion_num_frames, 'dt': ml._motion_dt,
                   'length_starts': ml.length_starts, 'dof_body_ids': DOF_BODY_IDS, 'dof_offsets': DOF_OFFSETS,
                   'key_body_ids': KEY_BODY_IDS},
         'motion_ids': ids, 'times': t,
         'outputs': {k: v.clone() for k, v in zip(('root_pos', 'root_rot', 'dof_pos', 'root_vel', 'root_ang_vel', 'dof_vel', 'key_pos'), out)}}
    path = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'motion_state.pt')
    torch.save(G, path)
    print('wrote', path, {k: tuple(v.shape) for k, v in G['outputs'].items()}, '%.1f KB' % (os.path.getsize(path) / 1e3))


if __name__ == '__main__':
    main()
