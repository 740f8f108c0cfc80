"""Golden vectors of the demo-side AMP-observation pipeline: the reference's OWN HumanoidAMP.fetch_amp_obs_demo /
build_amp_obs_demo (env/tasks/humanoid_amp.py:63-105, unbound, on a stub task object) over its OWN MotionLib
(utils/motion_lib.py) loaded from two shipped clips - sample_motions + sample_time under torch.manual_seed, the negative
time offsets, get_motion_state, build_amp_observations.  Quaternion primitives: the isaacgym restatement of
oracle/rl_games_shim.  The clip arrays themselves are those of tests/golden/motion_state.pt (same clips, same loader).
    python oracle/make_golden_demo.py  ->  tests/golden/amp_obs_demo.pt"""
import os
import sys
import tempfile
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'rl_games_shim'))
sys.path.insert(0, '/root/reference/ase')

from env.tasks import humanoid_amp as H       # noqa: E402  (reference code)
from utils.motion_lib import MotionLib        # noqa: E402  (reference code)

from make_golden_motion import CLIPS, DOF_BODY_IDS, DOF_OFFSETS, KEY_BODY_IDS   # noqa: E402

SEED, N, STEPS, DT = 2024, 64, 10, 1.0 / 30.0


def main():
    d = '/root/reference/ase/data/motions/reallusion_sword_shield'
    with tempfile.TemporaryDirectory() as tmp:
        y = os.path.join(tmp, 'two.yaml')
        with open(y, 'w') as f:
            f.write('motions:\n' + ''.join(f'  - file: "{os.path.join(d, c)}"\n    weight: {w}\n' for c, w in zip(CLIPS, (0.3, 0.7))))
        ml = MotionLib(motion_file=y, dof_body_ids=DOF_BODY_IDS, dof_offsets=DOF_OFFSETS, key_body_ids=KEY_BODY_IDS, device='cpu')
    G = {'seed': SEED, 'n': N, 'steps': STEPS, 'dt': DT, 'weights': ml._motion_weights.clone(), 'cases': {}}
    for local_root, root_h in ((True, True), (False, False)):
        task = types.SimpleNamespace(_motion_lib=ml, dt=DT, _num_amp_obs_steps=STEPS, _num_amp_obs_per_step=140, device='cpu',
                                     _local_root_obs=local_root, _root_height_obs=root_h, _dof_obs_size=6 * (len(DOF_OFFSETS) - 1),
                                     _dof_offsets=DOF_OFFSETS, _amp_obs_demo_buf=None)
        for name in ('fetch_amp_obs_demo', 'build_amp_obs_demo', '_build_amp_obs_demo_buf', 'get_num_amp_obs'):
            setattr(task, name, types.MethodType(getattr(H.HumanoidAMP, name), task))
        # record what the sampler drew (same seed, same call order as inside fetch_amp_obs_demo)
        torch.manual_seed(SEED)
        ids = ml.sample_motions(N)
        t0 = ml.sample_time(ids, truncate_time=DT * (STEPS - 1)) + DT * (STEPS - 1)
        torch.manual_seed(SEED)
        out = task.fetch_amp_obs_demo(N).clone()
        G['cases'][(local_root, root_h)] = {'motion_ids': ids, 'motion_times0': t0, 'amp_obs_demo': out}
    path = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'amp_obs_demo.pt')
    torch.save(G, path)
    print('wrote', path, tuple(out.shape), '%.1f KB' % (os.path.getsize(path) / 1e3))


if __name__ == '__main__':
    main()
