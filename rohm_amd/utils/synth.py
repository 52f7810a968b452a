"""Deterministic synthetic weights / body model / inputs.

No RoHM checkpoint, SMPL-X model file or AMASS clip exists in this environment
(SURVEY.md §8(c)), so benchmarks, smoke tests and parity tests all run on
synthetic tensors that have the *shapes and state_dict keys* of the released
artefacts.  Everything here is seeded through numpy's PCG64 so the same seed gives
the same bytes on the build container and on the GPU box.

State-dict key families follow the reference modules
(`model/posenet.py:59-72`, `model/trajnet.py:17-41,120-174`, `model/heads.py`).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch

# SMPL-X kinematic tree (55 joints), smplx==0.1.28 `parents` buffer.
SMPLX_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19,
                 15, 15, 15, 20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
                 21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53]


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def _u(rng, shape, bound):
    return torch.from_numpy(rng.uniform(-bound, bound, size=shape).astype(np.float32))


def _n(rng, shape, std=1.0):
    return torch.from_numpy((rng.standard_normal(size=shape) * std).astype(np.float32))


def sinusoid_table(d_model, max_len=5000):
    """`PositionalEncoding.pe` buffer, `model/heads.py:117-124` (shape [max_len, 1, d])."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-np.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(1)


def posenet_state_dict(seed=0, latent_dim=512, ff_size=1024, num_layers=8, body_feat_dim=294,
                       pose_feat_dim=272):
    """Random PoseNet weights with the 108 reference keys (no `smplx_model.*`)."""
    rng = _rng(seed)
    d = latent_dim
    sd = OrderedDict()

    def lin(prefix, n_out, n_in, gain=1.0):
        sd[prefix + '.weight'] = _u(rng, (n_out, n_in), gain * math.sqrt(3.0 / n_in))
        sd[prefix + '.bias'] = _u(rng, (n_out,), 0.1)

    lin('input_process.poseEmbedding', d, body_feat_dim)
    lin('input_process_cond.poseEmbedding', d, body_feat_dim)
    sd['sequence_pos_encoder.pe'] = sinusoid_table(d)
    for i in range(num_layers):
        p = f'seqTransEncoder.layers.{i}.'
        sd[p + 'self_attn.in_proj_weight'] = _u(rng, (3 * d, d), 1.2 * math.sqrt(3.0 / d))
        sd[p + 'self_attn.in_proj_bias'] = _u(rng, (3 * d,), 0.1)
        lin(p + 'self_attn.out_proj', d, d)
        lin(p + 'linear1', ff_size, d)
        lin(p + 'linear2', d, ff_size)
        for nm in ('norm1', 'norm2'):
            sd[p + nm + '.weight'] = 1.0 + _u(rng, (d,), 0.2)
            sd[p + nm + '.bias'] = _u(rng, (d,), 0.1)
    sd['embed_timestep.sequence_pos_encoder.pe'] = sd['sequence_pos_encoder.pe']
    lin('embed_timestep.time_embed.0', d, d)
    lin('embed_timestep.time_embed.2', d, d)
    lin('output_process.poseFinal', pose_feat_dim, d)
    return sd


def _res_block(sd, rng, prefix, c_in, c_out, input_t, t_dim=32, k=5):
    """Keys of one `ResidualTemporalBlock` (`model/heads.py:20-54`)."""
    for bi, ci in ((0, c_in), (1, c_out)):
        b = f'{prefix}.blocks.{bi}.block.'
        sd[b + '0.weight'] = _u(rng, (c_out, ci, k), math.sqrt(3.0 / (ci * k)))
        sd[b + '0.bias'] = _u(rng, (c_out,), 0.1)
        sd[b + '2.weight'] = 1.0 + _u(rng, (c_out,), 0.2)
        sd[b + '2.bias'] = _u(rng, (c_out,), 0.1)
    if input_t:
        sd[prefix + '.time_mlp.1.weight'] = _u(rng, (c_out, t_dim), math.sqrt(3.0 / t_dim))
        sd[prefix + '.time_mlp.1.bias'] = _u(rng, (c_out,), 0.1)
    if c_in != c_out:
        sd[prefix + '.residual_conv.weight'] = _u(rng, (c_out, c_in, 1), math.sqrt(3.0 / c_in))
        sd[prefix + '.residual_conv.bias'] = _u(rng, (c_out,), 0.1)


def _conv(sd, rng, name, c_out, c_in, k, scale=1.0):
    sd[name + '.weight'] = _u(rng, (c_out, c_in, k), scale * math.sqrt(3.0 / (c_in * k)))
    sd[name + '.bias'] = _u(rng, (c_out,), 0.1 * scale)


def trajnet_state_dict(seed=0, mid_dim=512, time_dim=32, traj_feat_dim=13, cond_dim=13,
                       trajcontrol=False, control_cond_dim=272, zero_convs_random=True):
    """Random TrajNet (+ControlNet) weights with the reference's 186 (+84) keys.

    The reference zero-initialises the six control 1x1 convs (`model/heads.py:12-18`);
    `zero_convs_random=True` randomises them (small scale) so the control branch is
    actually exercised by parity tests.
    """
    rng = _rng(seed)
    m = mid_dim
    sd = OrderedDict()
    if trajcontrol:
        c = 'controlnet.'
        zs = 0.3 if zero_convs_random else 0.0
        _conv(sd, rng, c + 'control_zero_conv_0', traj_feat_dim, control_cond_dim, 1, zs)
        _res_block(sd, rng, c + 'control_enc1', traj_feat_dim, m // 8, True, time_dim)
        _conv(sd, rng, c + 'control_zero_conv_1', 32, m // 8, 1, zs)
        _conv(sd, rng, c + 'control_downsample1.conv', m // 4, m // 4, 3)
        _res_block(sd, rng, c + 'control_enc2', m // 4, m // 4, True, time_dim)
        _conv(sd, rng, c + 'control_zero_conv_2', m // 8, m // 4, 1, zs)
        _conv(sd, rng, c + 'control_downsample2.conv', m // 2, m // 2, 3)
        _res_block(sd, rng, c + 'control_enc3', m // 2, m // 2, True, time_dim)
        _conv(sd, rng, c + 'control_zero_conv_3', m // 4, m // 2, 1, zs)
        _conv(sd, rng, c + 'control_downsample3.conv', m, m, 3)
        _res_block(sd, rng, c + 'control_enc4', m, m, True, time_dim)
        _conv(sd, rng, c + 'control_zero_conv_4', m // 2, m, 1, zs)
        _conv(sd, rng, c + 'control_downsample4.conv', 2 * m, 2 * m, 3)
        _res_block(sd, rng, c + 'control_mid_block1', 2 * m, m, True, time_dim)
        _res_block(sd, rng, c + 'control_mid_block2', m, m, True, time_dim)
        _conv(sd, rng, c + 'control_zero_conv_mid', m, m, 1, zs)
    sd['time_mlp.1.weight'] = _u(rng, (4 * time_dim, time_dim), math.sqrt(3.0 / time_dim))
    sd['time_mlp.1.bias'] = _u(rng, (4 * time_dim,), 0.1)
    sd['time_mlp.3.weight'] = _u(rng, (time_dim, 4 * time_dim), math.sqrt(3.0 / (4 * time_dim)))
    sd['time_mlp.3.bias'] = _u(rng, (time_dim,), 0.1)
    _res_block(sd, rng, 'diff_enc1', traj_feat_dim, m // 8, True, time_dim)
    _conv(sd, rng, 'diff_downsample1.conv', m // 4, m // 4, 3)
    _res_block(sd, rng, 'diff_enc2', m // 4, m // 4, True, time_dim)
    _conv(sd, rng, 'diff_downsample2.conv', m // 2, m // 2, 3)
    _res_block(sd, rng, 'diff_enc3', m // 2, m // 2, True, time_dim)
    _conv(sd, rng, 'diff_downsample3.conv', m, m, 3)
    _res_block(sd, rng, 'diff_enc4', m, m, True, time_dim)
    _conv(sd, rng, 'diff_downsample4.conv', 2 * m, 2 * m, 3)
    _res_block(sd, rng, 'diff_mid_block1', 2 * m, m, True, time_dim)
    _res_block(sd, rng, 'diff_mid_block2', m, m, True, time_dim)
    # ConvTranspose1d weight is [C_in, C_out, k] (`model/heads.py:84`)
    _conv(sd, rng, 'diff_upsample4.conv', m, m, 4)
    _res_block(sd, rng, 'diff_dec4', 2 * m, m // 2, True, time_dim)
    _conv(sd, rng, 'diff_upsample3.conv', m // 2, m // 2, 4)
    _res_block(sd, rng, 'diff_dec3', m, m // 4, True, time_dim)
    _conv(sd, rng, 'diff_upsample2.conv', m // 4, m // 4, 4)
    _res_block(sd, rng, 'diff_dec2', m // 2, m // 8, True, time_dim)
    _conv(sd, rng, 'diff_upsample1.conv', m // 8, m // 8, 4)
    _res_block(sd, rng, 'diff_dec1', m // 4, 32, True, time_dim)
    b = 'diff_final_conv.0.block.'
    sd[b + '0.weight'] = _u(rng, (32, 32, 5), math.sqrt(3.0 / 160))
    sd[b + '0.bias'] = _u(rng, (32,), 0.1)
    sd[b + '2.weight'] = 1.0 + _u(rng, (32,), 0.2)
    sd[b + '2.bias'] = _u(rng, (32,), 0.1)
    _conv(sd, rng, 'diff_final_conv.1', traj_feat_dim, 32, 1)
    _res_block(sd, rng, 'cond_enc1', cond_dim, m // 8, False)
    _conv(sd, rng, 'cond_downsample1.conv', m // 8, m // 8, 3)
    _res_block(sd, rng, 'cond_enc2', m // 8, m // 4, False)
    _conv(sd, rng, 'cond_downsample2.conv', m // 4, m // 4, 3)
    _res_block(sd, rng, 'cond_enc3', m // 4, m // 2, False)
    _conv(sd, rng, 'cond_downsample3.conv', m // 2, m // 2, 3)
    _res_block(sd, rng, 'cond_enc4', m // 2, m, False)
    _conv(sd, rng, 'cond_downsample4.conv', m, m, 3)  # built but never called (trajnet.py:174)
    return sd


def synthetic_smplx_tensors(seed=0, num_verts=10475, num_joints=55, num_betas=10, num_expr=10):
    """Synthetic SMPL-X model tensors with the real shapes.

    A crude humanoid: joints laid out along the kinematic tree, vertices scattered
    around their dominant joint, row-stochastic J_regressor, row-softmax skinning
    weights, small shape/pose blendshapes.
    """
    rng = _rng(seed)
    J, V = num_joints, num_verts
    parents = SMPLX_PARENTS[:J]
    # rest joints: pelvis ~1 m up (z), children offset 8-25 cm from their parent
    jpos = np.zeros((J, 3), np.float64)
    jpos[0] = (0.0, 0.0, 0.95)
    for j in range(1, J):
        step = rng.normal(size=3)
        step /= np.linalg.norm(step)
        jpos[j] = jpos[parents[j]] + step * rng.uniform(0.08, 0.25)
    owner = rng.integers(0, J, size=V)
    v_template = jpos[owner] + rng.normal(size=(V, 3)) * 0.04
    shapedirs = rng.normal(size=(V, 3, num_betas + num_expr)) * 0.01
    posedirs = rng.normal(size=((J - 1) * 9, V * 3)) * 0.002
    jr = rng.uniform(size=(J, V)) ** 8
    for j in range(J):
        jr[j] *= (owner == j) + 0.002
    jr /= jr.sum(1, keepdims=True)
    logits = rng.normal(size=(V, J)) * 1.5
    logits[np.arange(V), owner] += 4.0
    w = np.exp(logits - logits.max(1, keepdims=True))
    w /= w.sum(1, keepdims=True)
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return {
        'v_template': f32(v_template), 'shapedirs': f32(shapedirs), 'posedirs': f32(posedirs),
        'J_regressor': f32(jr), 'lbs_weights': f32(w),
        'parents': torch.tensor(parents, dtype=torch.long),
    }


def synthetic_stats(seed=0, dim=294):
    """Dataset `Mean`/`Std` stand-ins (np.float32 [dim]) -- `dataloader_amass.py:264-276`."""
    rng = _rng(seed + 7919)
    mean = (rng.standard_normal(dim) * 0.05).astype(np.float32)
    std = rng.uniform(0.5, 1.5, size=dim).astype(np.float32)
    return mean, std


def _rodrigues_np(aa):
    """Axis-angle [N,3] -> rotation matrices [N,3,3] (float64 numpy)."""
    aa = np.asarray(aa, dtype=np.float64)
    ang = np.linalg.norm(aa, axis=1, keepdims=True)
    ax = aa / np.maximum(ang, 1e-12)
    K = np.zeros((aa.shape[0], 3, 3))
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0] = -ax[:, 2], ax[:, 1], ax[:, 2]
    K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -ax[:, 0], -ax[:, 1], ax[:, 0]
    s, c = np.sin(ang)[:, :, None], np.cos(ang)[:, :, None]
    return np.eye(3)[None] + s * K + (1 - c) * (K @ K)


def plausible_motion(seed, B, T, mean, std, angle_scale=0.4, trans_z=3.0):
    """Normalised x0 [B, 294, 1, T] whose de-normalised content is a plausible body: joint rotations of
    ~`angle_scale` rad written as interleaved 6-D vectors (first two columns of R, as get_repr_smplx stores
    them, motion_representation.py:248,261), the body ~`trans_z` m along +z (in front of a camera), contact
    labels well away from the 0.5 threshold, everything else small Gaussian."""
    g = _rng(seed)
    full = (g.standard_normal((B, T, 294)) * 0.3).astype(np.float32)

    def six_d(aa):
        R = _rodrigues_np(aa.reshape(-1, 3))
        return R[:, :, :2].reshape(-1, 6).astype(np.float32)
    full[..., 7:13] = six_d(g.standard_normal((B, T, 3)) * angle_scale).reshape(B, T, 6)
    full[..., 154:280] = six_d(g.standard_normal((B, T, 21, 3)) * angle_scale).reshape(B, T, 126)
    full[..., 16:19] = np.stack([g.standard_normal((B, T)) * 0.3, g.standard_normal((B, T)) * 0.3,
                                 trans_z + 0.2 * g.standard_normal((B, T))], -1)
    full[..., 290:294] = (g.uniform(size=(B, T, 4)) > 0.5).astype(np.float32) * 0.9 + 0.05
    x = (full - mean) / std
    return torch.from_numpy(x.astype(np.float32)).permute(0, 2, 1).unsqueeze(2).contiguous()


def walking_motion(seed, B, T, mean, std, body_tensors, yaw_range=0.6, pose_scale=0.12):
    """Normalised [B, T, 294] clips of a SMOOTH motion whose facing direction is well conditioned: the global
    orientation is a slow yaw (|yaw| <= `yaw_range` rad) on top of the rotation that turns the rest body's
    hip+shoulder axis onto +x (so `get_repr_smplx`'s forward direction stays near +y, away from the -y
    singularity of its quaternion, motion_representation.py:199-211), joint angles are slow sinusoids of
    ~`pose_scale` rad, the root drifts ~1 m.  For driver-level (multi-stage) parity tests, where a frame with an
    ill-conditioned facing direction would dominate the comparison."""
    g = _rng(seed)
    jr = body_tensors['J_regressor'].double().numpy() @ body_tensors['v_template'].double().numpy()
    a = (jr[1] - jr[2]) + (jr[17] - jr[16])
    a /= np.linalg.norm(a)
    ax = np.cross(a, [1.0, 0.0, 0.0])
    sn, cs = np.linalg.norm(ax), float(a[0])
    R0 = _rodrigues_np((ax / max(sn, 1e-12) * np.arctan2(sn, cs))[None])[0]
    tt = np.linspace(0.0, 1.0, T)[None, :]
    full = (g.standard_normal((B, T, 294)) * 0.05).astype(np.float32)

    def slow(shape_tail, amp):
        # a couple of low-frequency sinusoids with random phase per clip / component
        f = g.uniform(0.3, 1.5, size=(B, 1) + shape_tail)
        ph = g.uniform(0, 2 * np.pi, size=(B, 1) + shape_tail)
        return amp * np.sin(2 * np.pi * f * tt.reshape((1, T) + (1,) * len(shape_tail)) + ph)
    yaw = slow((), yaw_range)                                               # [B, T]
    Rz = np.zeros((B, T, 3, 3))
    Rz[..., 0, 0], Rz[..., 0, 1], Rz[..., 1, 0], Rz[..., 1, 1], Rz[..., 2, 2] = np.cos(yaw), -np.sin(yaw), np.sin(yaw), np.cos(yaw), 1.0
    Rg = Rz @ R0
    full[..., 7:13] = Rg[..., :, :2].reshape(B, T, 6).astype(np.float32)
    Rb = _rodrigues_np(slow((21, 3), pose_scale).reshape(-1, 3))
    full[..., 154:280] = Rb[:, :, :2].reshape(B, T, 126).astype(np.float32)
    full[..., 16:19] = np.stack([slow((), 0.5), slow((), 0.5), 0.1 * slow((), 1.0)], -1).astype(np.float32)
    full[..., 280:290] = (g.standard_normal((B, 1, 10)) * 0.3).astype(np.float32)
    full[..., 290:294] = (g.uniform(size=(B, T, 4)) > 0.5).astype(np.float32) * 0.9 + 0.05
    return torch.from_numpy(((full - mean) / std).astype(np.float32))


def synthetic_camera_batch(seed, B, frames=145):
    """PROX-like guidance inputs (SURVEY.md §8d cfg 4): Kinect-colour intrinsics
    (utils/get_occlusion_mask.py:64-68), rigid cano->scene transforms, OpenPose-style keypoints."""
    g = _rng(seed + 31)
    tm = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    tm[:, :3, :3] = _rodrigues_np(g.standard_normal((B, 3)) * 0.2).astype(np.float32)
    tm[:, :3, 3] = (g.standard_normal((B, 3)) * 0.2).astype(np.float32)
    kp = np.concatenate([g.uniform(size=(B, frames, 22, 2)) * np.array([1920., 1080.]),
                         g.uniform(size=(B, frames, 22, 1))], -1).astype(np.float32)
    return {
        'transf_matrix': torch.from_numpy(tm),
        'focal_length': torch.tensor([[1060.53, 1060.38]], dtype=torch.float32).repeat(B, 1),
        'camera_center': torch.tensor([[951.30, 536.77]], dtype=torch.float32).repeat(B, 1),
        'keypoints_2d': torch.from_numpy(kp),
    }


SYNTH_CAM_R = [[0.99, 0.1, 0.05], [-0.1, 0.98, 0.02], [-0.04, -0.03, 1.0]]
SYNTH_CAM_T = [[0.1, -0.2, 0.3]]
