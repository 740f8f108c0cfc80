"""Demo-side AMP-observation production on the device: the sampling / state interface of the reference's
``MotionLib`` (utils/motion_lib.py:100-172 under /root/reference/ase) and ``HumanoidAMP.fetch_amp_obs_demo`` /
``build_amp_obs_demo`` (env/tasks/humanoid_amp.py:63-105) over clip arrays resident in HBM.

The clip arrays (per-frame global translations / rotations, local rotations, root velocities, dof velocities of all clips
concatenated, plus per-clip length / frame count / dt / first frame) are what the reference's loader
(``MotionLib._load_motions``: poselib ``SkeletonMotion.from_file`` + ``_compute_motion_dof_vels``) leaves in
``ml.gts, ml.grs, ml.lrs, ml.grvs, ml.gravs, ml.dvs`` - the loader stays reference code (it runs once, on the host);
``DeviceMotionLib.from_reference`` takes such an object, ``from_arrays`` the arrays themselves.  Everything per sample -
frame blend, slerp of the root and every local rotation, exponential-map dof positions, key-body interpolation, then the
140-float observation frame (root height, tangent-normal root rotation, local velocities, dof observations, key bodies) -
runs in two HIP kernels (``ase_hip_motion_state``, ``ase_hip_build_amp_obs``); there is no host fallback.
"""
import torch


class DeviceMotionLib:
    """``MotionLib`` (utils/motion_lib.py:57) without the loader: same method names, argument meaning and return order."""

    CLIP_F32 = ('gts', 'grs', 'lrs', 'grvs', 'gravs', 'dvs', 'lengths', 'dt')
    CLIP_I32 = ('num_frames', 'length_starts')

    def __init__(self, clips, backend, device, weights=None, generator=None):
        self.be, self._device, self.gen = backend, torch.device(device), generator
        self.clips = {k: clips[k].to(torch.float32).contiguous().to(self._device) for k in self.CLIP_F32}
        self.clips.update({k: clips[k].to(torch.int32).contiguous().to(self._device) for k in self.CLIP_I32})
        self.clips.update({k: [int(x) for x in clips[k]] for k in ('dof_body_ids', 'dof_offsets', 'key_body_ids')})
        self._motion_lengths = self.clips['lengths']
        n = self._motion_lengths.shape[0]
        w = torch.ones(n) if weights is None else torch.as_tensor(weights, dtype=torch.float32)
        self._motion_weights = (w / w.sum()).to(self._device)                 # motion_lib.py:213

    @classmethod
    def from_arrays(cls, clips, backend, device, **kw):
        return cls(clips, backend, device, **kw)

    @classmethod
    def from_reference(cls, ml, backend, device, dof_body_ids, dof_offsets, key_body_ids, **kw):
        """From a loaded reference ``MotionLib`` (any device): copies its frame arrays and per-clip tables."""
        clips = {'gts': ml.gts, 'grs': ml.grs, 'lrs': ml.lrs, 'grvs': ml.grvs, 'gravs': ml.gravs, 'dvs': ml.dvs,
                 'lengths': ml._motion_lengths, 'num_frames': ml._motion_num_frames, 'dt': ml._motion_dt,
                 'length_starts': ml.length_starts, 'dof_body_ids': dof_body_ids, 'dof_offsets': dof_offsets,
                 'key_body_ids': key_body_ids}
        kw.setdefault('weights', ml._motion_weights)
        return cls(clips, backend, device, **kw)

    def num_motions(self):
        return int(self._motion_lengths.shape[0])

    def get_total_length(self):
        return float(self._motion_lengths.sum())

    def get_motion_length(self, motion_ids):
        return self._motion_lengths[motion_ids]

    def sample_motions(self, n):
        """motion_lib.py:100-106."""
        return torch.multinomial(self._motion_weights, num_samples=n, replacement=True, generator=self.gen)

    def sample_time(self, motion_ids, truncate_time=None):
        """motion_lib.py:108-119: uniform phase times the (truncated) clip length."""
        phase = torch.rand(motion_ids.shape, device=self._device, generator=self.gen)
        motion_len = self._motion_lengths[motion_ids]
        if truncate_time is not None:
            assert truncate_time >= 0.0
            motion_len = motion_len - truncate_time
        return phase * motion_len

    def get_motion_state(self, motion_ids, motion_times):
        """motion_lib.py:122-172 -> (root_pos, root_rot, dof_pos, root_vel, root_ang_vel, dof_vel, key_pos)."""
        ids = motion_ids.to(torch.int32).contiguous()
        return self.be.motion_state(self.clips, ids, motion_times.to(torch.float32).contiguous())


class AmpObsDemoSource:
    """``HumanoidAMP.fetch_amp_obs_demo`` (env/tasks/humanoid_amp.py:63-84): ``num_samples`` demo observations of
    ``num_amp_obs_steps`` frames each, newest frame first, frame k taken ``k * dt`` before the sampled time."""

    def __init__(self, motion_lib, backend, num_amp_obs_steps=10, dt=1.0 / 30.0, local_root_obs=True, root_height_obs=True):
        self._motion_lib, self.be = motion_lib, backend
        self._num_amp_obs_steps, self.dt = int(num_amp_obs_steps), float(dt)
        self._local_root_obs, self._root_height_obs = bool(local_root_obs), bool(root_height_obs)
        c = motion_lib.clips
        n_joints = len(c['dof_offsets']) - 1
        self._num_amp_obs_per_step = 13 + 6 * n_joints + c['dof_offsets'][-1] + 3 * len(c['key_body_ids'])   # humanoid_amp.py:107-118
        self._amp_obs_demo_buf = None

    def get_num_amp_obs(self):
        return self._num_amp_obs_steps * self._num_amp_obs_per_step

    def fetch_amp_obs_demo(self, num_samples):
        ml = self._motion_lib
        motion_ids = ml.sample_motions(num_samples)
        # negative offsets are added in build_amp_obs_demo: shift the times into [truncate_time, end of clip]
        truncate_time = self.dt * (self._num_amp_obs_steps - 1)
        motion_times0 = ml.sample_time(motion_ids, truncate_time=truncate_time)
        motion_times0 = motion_times0 + truncate_time
        return self.build_amp_obs_demo(motion_ids, motion_times0).view(num_samples, self.get_num_amp_obs())

    def build_amp_obs_demo(self, motion_ids, motion_times0):
        """humanoid_amp.py:86-101 -> [n, steps, per_step] (a fresh buffer per call size, reused between calls)."""
        n, S = motion_ids.shape[0], self._num_amp_obs_steps
        dev = motion_times0.device
        ids = motion_ids.view(-1, 1).expand(n, S).reshape(-1)
        time_steps = -self.dt * torch.arange(0, S, device=dev)
        times = (motion_times0.unsqueeze(-1) + time_steps).reshape(-1)
        root_pos, root_rot, dof_pos, root_vel, root_ang_vel, dof_vel, key_pos = self._motion_lib.get_motion_state(ids, times)
        if self._amp_obs_demo_buf is None or self._amp_obs_demo_buf.shape[0] != n * S:
            self._amp_obs_demo_buf = torch.zeros(n * S, 1, self._num_amp_obs_per_step, device=dev, dtype=torch.float32)
        self.be.build_amp_obs(root_pos, root_rot, root_vel, root_ang_vel, dof_pos, dof_vel, key_pos,
                              self._motion_lib.clips['dof_offsets'], self._local_root_obs, self._root_height_obs,
                              self._amp_obs_demo_buf, shift=False)
        return self._amp_obs_demo_buf.view(n, S, self._num_amp_obs_per_step)
