"""CPU restatement of the AMP-observation builder.  TEST INFRASTRUCTURE ONLY.

Follows env/tasks/humanoid_amp.py:280-316 (``build_amp_observations``), env/tasks/humanoid.py:523-552
(``dof_to_obs``), utils/torch_utils.py:51-63,68-91,150-190 (tan-norm rotation, exponential map, heading) and
env/tasks/humanoid_amp.py:248-256 (``_update_hist_amp_obs``) under /root/reference/ase, with the quaternion algebra
written out explicitly (xyzw, Hamilton product).  The algebra the reference takes from ``isaacgym.torch_utils`` is
not vendored by the reference: **parity unpinned at the isaacgym boundary** (oracle/rl_games_shim/isaacgym restates
it; tests/golden/amp_obs.pt = the reference's own function running on that restatement, oracle/make_golden_amp_obs.py).
"""
import torch


def quat_rotate(q, v):
    """Rotate v [n, 3] by unit quaternions q [n, 4] (xyzw)."""
    w, u = q[:, 3:4], q[:, :3]
    return v * (2.0 * w * w - 1.0) + torch.cross(u, v, dim=-1) * w * 2.0 + u * (u * v).sum(-1, keepdim=True) * 2.0


def quat_mul(a, b):
    x1, y1, z1, w1 = a.unbind(-1)
    x2, y2, z2, w2 = b.unbind(-1)
    return torch.stack([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2,
                        w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2], -1)


def quat_from_angle_axis(angle, axis):
    axis = axis / axis.norm(dim=-1, keepdim=True).clamp_min(1e-9)
    h = angle.unsqueeze(-1) * 0.5
    q = torch.cat([axis * torch.sin(h), torch.cos(h)], -1)
    return q / q.norm(dim=-1, keepdim=True).clamp_min(1e-9)


def quat_to_tan_norm(q):
    """utils/torch_utils.py:51-63: the rotated x axis (tangent) and z axis (normal), 6 numbers."""
    ex = torch.zeros_like(q[:, :3]); ex[:, 0] = 1
    ez = torch.zeros_like(q[:, :3]); ez[:, 2] = 1
    return torch.cat([quat_rotate(q, ex), quat_rotate(q, ez)], -1)


def heading_quat_inv(q):
    """utils/torch_utils.py:150-190: rotation about z by minus the heading of q."""
    ex = torch.zeros_like(q[:, :3]); ex[:, 0] = 1
    d = quat_rotate(q, ex)
    heading = torch.atan2(d[:, 1], d[:, 0])
    ez = torch.zeros_like(q[:, :3]); ez[:, 2] = 1
    return quat_from_angle_axis(-heading, ez)


def exp_map_to_quat(e):
    """utils/torch_utils.py:68-91: angle = |e| wrapped to (-pi, pi], axis = e / |e| (z axis and angle 0 below 1e-5)."""
    angle = e.norm(dim=-1)
    axis = e / angle.unsqueeze(-1)
    angle = torch.atan2(torch.sin(angle), torch.cos(angle))
    ok = angle.abs() > 1e-5
    ez = torch.zeros_like(e); ez[:, 2] = 1
    angle = torch.where(ok, angle, torch.zeros_like(angle))
    axis = torch.where(ok.unsqueeze(-1), axis, ez)
    return quat_from_angle_axis(angle, axis)


def dof_to_obs(pose, dof_offsets):
    """env/tasks/humanoid.py:523-552: 3-dof joints are exponential maps, 1-dof joints rotate about y; 6 numbers each."""
    out = []
    for j in range(len(dof_offsets) - 1):
        a, b = dof_offsets[j], dof_offsets[j + 1]
        if b - a == 3:
            q = exp_map_to_quat(pose[:, a:b])
        elif b - a == 1:
            ey = torch.zeros(pose.shape[0], 3, dtype=pose.dtype); ey[:, 1] = 1
            q = quat_from_angle_axis(pose[:, a], ey)
        else:
            raise ValueError('unsupported joint type')
        out.append(quat_to_tan_norm(q))
    return torch.cat(out, -1)


def build_amp_observations(root_pos, root_rot, root_vel, root_ang_vel, dof_pos, dof_vel, key_body_pos, local_root_obs,
                           root_height_obs, dof_offsets):
    """env/tasks/humanoid_amp.py:280-316 -> [n, 1 + 6 + 3 + 3 + 6 J + D + 3 K]."""
    n = root_pos.shape[0]
    hq = heading_quat_inv(root_rot)
    rr = quat_mul(hq, root_rot) if local_root_obs else root_rot
    h = root_pos[:, 2:3] if root_height_obs else torch.zeros_like(root_pos[:, 2:3])
    K = key_body_pos.shape[1]
    local_key = key_body_pos - root_pos.unsqueeze(1)
    hq_k = hq.unsqueeze(1).expand(n, K, 4).reshape(n * K, 4)
    key = quat_rotate(hq_k, local_key.reshape(n * K, 3)).reshape(n, K * 3)
    return torch.cat([h, quat_to_tan_norm(rr), quat_rotate(hq, root_vel), quat_rotate(hq, root_ang_vel),
                      dof_to_obs(dof_pos, dof_offsets), dof_vel, key], -1)


def push_history(hist, frame):
    """env/tasks/humanoid_amp.py:248-256 + 258-266: slots shift towards the past, the new frame takes slot 0."""
    hist[:, 1:] = hist[:, :-1].clone()
    hist[:, 0] = frame
    return hist


# ---------------------------------------------------------------------------------------------------------------------
# Motion clip sampler (utils/motion_lib.py:122-172 get_motion_state, :263-272 _calc_frame_blend, :296-325
# _local_rotation_to_dof; utils/torch_utils.py:7-28 quat_to_angle_axis, :94-118 slerp)
# ---------------------------------------------------------------------------------------------------------------------
def slerp(q0, q1, t):
    c = (q0 * q1).sum(-1)
    q1 = torch.where((c < 0).unsqueeze(-1), -q1, q1)
    c = c.abs().unsqueeze(-1)
    ht = torch.acos(c)
    s = torch.sqrt(1.0 - c * c)
    out = torch.sin((1 - t) * ht) / s * q0 + torch.sin(t * ht) / s * q1
    out = torch.where(s.abs() < 0.001, 0.5 * q0 + 0.5 * q1, out)
    return torch.where(c.abs() >= 1, q0, out)


def quat_to_angle_axis(q):
    w = q[..., 3]
    s = torch.sqrt(1 - w * w)
    angle = 2 * torch.acos(w)
    angle = torch.atan2(torch.sin(angle), torch.cos(angle))
    axis = q[..., :3] / s.unsqueeze(-1)
    ok = s.abs() > 1e-5
    ez = torch.zeros_like(axis); ez[..., 2] = 1
    return torch.where(ok, angle, torch.zeros_like(angle)), torch.where(ok.unsqueeze(-1), axis, ez)


def motion_state(clips, motion_ids, times):
    """clips: dict of the MotionLib tensors (gts [F, B, 3], grs / lrs [F, B, 4], grvs / gravs [F, 3], dvs [F, D],
    lengths / num_frames / dt / length_starts [n_motions], dof_body_ids, dof_offsets, key_body_ids)."""
    ln, nf, dt = clips['lengths'][motion_ids], clips['num_frames'][motion_ids], clips['dt'][motion_ids]
    phase = torch.clip(times / ln, 0.0, 1.0)
    i0 = (phase * (nf - 1)).long()
    i1 = torch.min(i0 + 1, nf - 1)
    blend = ((times - i0 * dt) / dt).unsqueeze(-1)
    f0, f1 = i0 + clips['length_starts'][motion_ids], i1 + clips['length_starts'][motion_ids]
    gts, grs, lrs = clips['gts'], clips['grs'], clips['lrs']
    root_pos = (1.0 - blend) * gts[f0, 0] + blend * gts[f1, 0]
    root_rot = slerp(grs[f0, 0], grs[f1, 0], blend)
    kb = torch.as_tensor(clips['key_body_ids'])
    key_pos = (1.0 - blend.unsqueeze(-1)) * gts[f0][:, kb] + blend.unsqueeze(-1) * gts[f1][:, kb]
    local_rot = slerp(lrs[f0], lrs[f1], blend.unsqueeze(-1))
    offs = clips['dof_offsets']
    dof_pos = torch.zeros(len(motion_ids), offs[-1])
    for j, body in enumerate(clips['dof_body_ids']):
        a, b = offs[j], offs[j + 1]
        angle, axis = quat_to_angle_axis(local_rot[:, body])
        if b - a == 3:
            dof_pos[:, a:b] = angle.unsqueeze(-1) * axis
        else:
            th = angle * axis[..., 1]
            dof_pos[:, a] = torch.atan2(torch.sin(th), torch.cos(th))
    return root_pos, root_rot, dof_pos, clips['grvs'][f0], clips['gravs'][f0], clips['dvs'][f0], key_pos
