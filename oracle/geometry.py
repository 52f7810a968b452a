"""CPU restatement of the geometry half of the hot path (test oracle, differentiable through torch autograd).

Restates, function by function:
  * `rot6d_to_rotmat`            data_loaders/common/quaternion.py:482-501
  * `qinv`, `qrot`               data_loaders/common/quaternion.py:14-18, 52-71
  * `rotation_matrix_to_quaternion`, `quaternion_to_angle_axis`, `rotation_matrix_to_angle_axis`
                                 utils/konia_transform.py:350-444, 561-631, 317-340 (+ :45-48, :344-347)
  * `recover_root_rot_pos`, `recover_from_repr_smpl`
                                 data_loaders/motion_representation.py:285-329, 332-398
  * `perspective_projection`     utils/other_utils.py:150-185
  * `guide_skating_with_smpl`, `guide_2d_projection_with_smpl`
                                 model/posenet.py:196-257, 260-317
  * SMPL-X forward (`SMPLX.forward`, `lbs`, `batch_rodrigues`, `batch_rigid_transform`,
    `blend_shapes`, `vertices2joints`) of the third-party package smplx==0.1.28
    (environment.yml:198) -- NOT vendored in the reference and not installed here, so this part
    follows the package's published algorithm and is PARITY UNPINNED (self-consistency tests only).
"""
from __future__ import annotations

import types

import torch
import torch.nn.functional as F

REPR_LIST = ['root_rot_angle', 'root_rot_angle_vel', 'root_l_pos', 'root_l_vel', 'root_height',
             'smplx_rot_6d', 'smplx_rot_vel', 'smplx_trans', 'smplx_trans_vel',
             'local_positions', 'local_vel', 'smplx_body_pose_6d', 'smplx_betas', 'foot_contact']
REPR_DIM = {'root_rot_angle': 1, 'root_rot_angle_vel': 1, 'root_l_pos': 2, 'root_l_vel': 2, 'root_height': 1,
            'smplx_rot_6d': 6, 'smplx_rot_vel': 3, 'smplx_trans': 3, 'smplx_trans_vel': 3,
            'local_positions': 66, 'local_vel': 66, 'smplx_body_pose_6d': 126, 'smplx_betas': 10,
            'foot_contact': 4}      # utils/other_utils.py:17-37 (294 channels)

FOOT_JOINTS = [7, 10, 8, 11]       # model/posenet.py:31
PROJ_JOINTS = [16, 18, 20, 17, 19, 21, 4, 5, 7, 8]   # model/posenet.py:308


def split_repr(full):
    """[..., 294] -> dict of named slices (model/posenet.py:213-215)."""
    out, o = {}, 0
    for name in REPR_LIST:
        out[name] = full[..., o:o + REPR_DIM[name]]
        o += REPR_DIM[name]
    return out


# ------------------------------------------------------------------------------ rotations
def rot6d_to_rotmat(x):
    x = x.reshape(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]                    # interleaved columns (quaternion.py:494-497)
    b1 = F.normalize(a1)
    b2 = F.normalize(a2 - torch.einsum('bi,bi->b', b1, a2).unsqueeze(-1) * b1)
    b3 = torch.cross(b1, b2, dim=1)
    return torch.stack((b1, b2, b3), dim=-1)


def qinv(q):
    mask = torch.ones_like(q)
    mask[..., 1:] = -mask[..., 1:]
    return q * mask


def qrot(q, v):
    shape = list(v.shape)
    q = q.contiguous().view(-1, 4)
    v = v.contiguous().view(-1, 3)
    qvec = q[:, 1:]
    uv = torch.cross(qvec, v, dim=1)
    uuv = torch.cross(qvec, uv, dim=1)
    return (v + 2 * (q[:, :1] * uv + uuv)).view(shape)


def _safe_zero_division(num, den, eps=1.0e-6):
    den = den.clone()
    den[den.abs() < eps] += eps
    return num / den


def _safe_atan2(y, x, eps=1e-6):
    y = y.clone()
    y[(y.abs() < eps) & (x.abs() < eps)] += eps
    return torch.atan2(y, x)


def rotation_matrix_to_quaternion(R, eps=1.0e-6):
    """wxyz quaternion, four-branch selection (konia_transform.py:350-444)."""
    v = R.reshape(*R.shape[:-2], 9)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.chunk(v, chunks=9, dim=-1)
    trace = m00 + m11 + m22

    def pos():
        sq = torch.sqrt((trace + 1.0).clamp_min(eps)) * 2.0
        return torch.cat((0.25 * sq, _safe_zero_division(m21 - m12, sq), _safe_zero_division(m02 - m20, sq),
                          _safe_zero_division(m10 - m01, sq)), dim=-1)

    def c1():
        sq = torch.sqrt((1.0 + m00 - m11 - m22).clamp_min(eps)) * 2.0
        return torch.cat((_safe_zero_division(m21 - m12, sq), 0.25 * sq, _safe_zero_division(m01 + m10, sq),
                          _safe_zero_division(m02 + m20, sq)), dim=-1)

    def c2():
        sq = torch.sqrt((1.0 + m11 - m00 - m22).clamp_min(eps)) * 2.0
        return torch.cat((_safe_zero_division(m02 - m20, sq), _safe_zero_division(m01 + m10, sq), 0.25 * sq,
                          _safe_zero_division(m12 + m21, sq)), dim=-1)

    def c3():
        sq = torch.sqrt((1.0 + m22 - m00 - m11).clamp_min(eps)) * 2.0
        return torch.cat((_safe_zero_division(m10 - m01, sq), _safe_zero_division(m02 + m20, sq),
                          _safe_zero_division(m12 + m21, sq), 0.25 * sq), dim=-1)

    w2 = torch.where(m11 > m22, c2(), c3())
    w1 = torch.where((m00 > m11) & (m00 > m22), c1(), w2)
    return torch.where(trace > 0.0, pos(), w1)


def quaternion_to_angle_axis(q, eps=1.0e-6):
    """konia_transform.py:561-631 (wxyz)."""
    cos_t, q1, q2, q3 = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    s2 = q1 * q1 + q2 * q2 + q3 * q3
    sin_t = torch.sqrt(s2.clamp_min(eps))
    two_theta = 2.0 * torch.where(cos_t < 0.0, _safe_atan2(-sin_t, -cos_t), _safe_atan2(sin_t, cos_t))
    k = torch.where(s2 > 0.0, _safe_zero_division(two_theta, sin_t, eps), 2.0 * torch.ones_like(sin_t))
    return torch.stack((q1 * k, q2 * k, q3 * k), dim=-1)


def rotation_matrix_to_angle_axis(R):
    return quaternion_to_angle_axis(rotation_matrix_to_quaternion(R))


# ------------------------------------------------------------------------------ SMPL-X (smplx==0.1.28)
def batch_rodrigues(rot_vecs):
    """smplx.lbs.batch_rodrigues: angle = ||r + 1e-8||, R = I + sin K + (1 - cos) K^2."""
    N = rot_vecs.shape[0]
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos, sin = torch.cos(angle)[:, None], torch.sin(angle)[:, None]
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros((N, 1), dtype=rot_vecs.dtype)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view(N, 3, 3)
    ident = torch.eye(3, dtype=rot_vecs.dtype)[None]
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


class BodyModel(torch.nn.Module):
    """SMPL-X body model (neutral, flat_hand_mean=True, use_pca=False, 10 betas + 10 expression coeffs) with
    the call signature the reference uses (motion_representation.py:379-396): returns an object with
    `.joints` [N, 127, 3] and `.vertices` [N, V, 3].  Only the first 55 joints (the kinematic tree) are the
    FK joints; smplx appends 72 vertex-picked / landmark joints, of which the hot path reads none
    (`joints[:, 0:22]`), so they are filled with zeros here."""

    def __init__(self, tensors, dtype=torch.float32):
        super().__init__()
        for k in ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'lbs_weights'):
            self.register_buffer(k, tensors[k].to(dtype))
        self.register_buffer('parents', tensors['parents'].clone())
        self.dtype = dtype

    def forward(self, betas, global_orient, body_pose, transl, jaw_pose=None, leye_pose=None, reye_pose=None,
                left_hand_pose=None, right_hand_pose=None, expression=None, return_verts=True, **kw):
        N = betas.shape[0]
        z = lambda n: torch.zeros(N, n, dtype=betas.dtype)
        jaw_pose = z(3) if jaw_pose is None else jaw_pose
        leye_pose = z(3) if leye_pose is None else leye_pose
        reye_pose = z(3) if reye_pose is None else reye_pose
        left_hand_pose = z(45) if left_hand_pose is None else left_hand_pose
        right_hand_pose = z(45) if right_hand_pose is None else right_hand_pose
        expression = z(10) if expression is None else expression
        full_pose = torch.cat([global_orient, body_pose, jaw_pose, leye_pose, reye_pose, left_hand_pose,
                               right_hand_pose], dim=1)                                # [N, 165]; pose_mean = 0
        shape_comp = torch.cat([betas, expression], dim=-1)                            # [N, 20]
        J_n = self.J_regressor.shape[0]
        v_shaped = self.v_template[None] + torch.einsum('bl,mkl->bmk', shape_comp, self.shapedirs)
        J = torch.einsum('bik,ji->bjk', v_shaped, self.J_regressor)                    # vertices2joints
        rot_mats = batch_rodrigues(full_pose.view(-1, 3)).view(N, -1, 3, 3)
        ident = torch.eye(3, dtype=betas.dtype)
        pose_feature = (rot_mats[:, 1:] - ident).view(N, -1)
        v_posed = v_shaped + torch.matmul(pose_feature, self.posedirs).view(N, -1, 3)
        # batch_rigid_transform
        parents = self.parents
        rel = J.clone()
        rel[:, 1:] = J[:, 1:] - J[:, parents[1:]]
        tm = torch.cat([torch.cat([rot_mats, rel[..., None]], dim=-1),
                        torch.tensor([0, 0, 0, 1], dtype=betas.dtype).expand(N, J_n, 1, 4)], dim=-2)
        chain = [tm[:, 0]]
        for i in range(1, J_n):
            chain.append(torch.matmul(chain[int(parents[i])], tm[:, i]))
        transforms = torch.stack(chain, dim=1)
        posed_joints = transforms[:, :, :3, 3]
        joints = torch.cat([posed_joints + transl[:, None], torch.zeros(N, 127 - J_n, 3, dtype=betas.dtype)], dim=1)
        verts = None
        if return_verts:
            Jh = torch.cat([J, torch.zeros(N, J_n, 1, dtype=betas.dtype)], dim=-1)[..., None]   # [N, J, 4, 1]
            rel_t = transforms - F.pad(torch.matmul(transforms, Jh), [3, 0, 0, 0, 0, 0, 0, 0])
            T = torch.matmul(self.lbs_weights[None].expand(N, -1, -1), rel_t.view(N, J_n, 16)).view(N, -1, 4, 4)
            vh = torch.cat([v_posed, torch.ones(N, v_posed.shape[1], 1, dtype=betas.dtype)], dim=2)
            verts = torch.matmul(T, vh[..., None])[:, :, :3, 0] + transl[:, None]
        return types.SimpleNamespace(joints=joints, vertices=verts)


# ------------------------------------------------------------------------------ recover_from_repr_smpl
def recover_root_rot_pos_abs(data):
    """'abs' branch with up_axis='z' (motion_representation.py:299-311): q = (cos a, 0, 0, sin a),
    r_pos = (x, y, height)."""
    ang = data[..., 0]
    q = torch.zeros(data.shape[:-1] + (4,), dtype=data.dtype)
    q[..., 0] = torch.cos(ang)
    q[..., 3] = torch.sin(ang)
    pos = torch.zeros(data.shape[:-1] + (3,), dtype=data.dtype)
    pos[..., [0, 1]] = data[..., 1:3]
    pos[..., 2] = data[..., 3]
    return q, pos


def joints_from_abs_traj(d):
    """recover_mode='joint_abs_traj' (motion_representation.py:349-371) -> [B, T, 22, 3]."""
    q, r_pos = recover_root_rot_pos_abs(torch.cat([d['root_rot_angle'], d['root_l_pos'], d['root_height']], -1))
    positions = d['local_positions'][..., 3:]
    positions = positions.reshape(positions.shape[:-1] + (-1, 3))
    positions = qrot(qinv(q[..., None, :]).expand(positions.shape[:-1] + (4,)), positions)
    positions = positions.clone()
    positions[..., 0] += r_pos[..., 0:1]
    positions[..., 1] += r_pos[..., 1:2]
    return torch.cat([r_pos.unsqueeze(-2), positions], dim=-2)


def recover_root_rot_pos_rel(data):
    """'rel' branch with up_axis='z' (motion_representation.py:312-329): the root angle is the running sum of the
    rotation velocities of the EARLIER frames, the root position the running sum of the earlier frames' velocities
    rotated back by qinv(q_t); height is absolute."""
    rot_vel = data[..., 0]
    ang = torch.zeros_like(rot_vel)
    ang[..., 1:] = rot_vel[..., :-1]
    ang = torch.cumsum(ang, dim=-1)
    q = torch.zeros(data.shape[:-1] + (4,), dtype=data.dtype)
    q[..., 0] = torch.cos(ang)
    q[..., 3] = torch.sin(ang)
    pos = torch.zeros(data.shape[:-1] + (3,), dtype=data.dtype)
    pos[..., 1:, [0, 1]] = data[..., :-1, 1:3]
    pos = qrot(qinv(q), pos)
    pos = torch.cumsum(pos, dim=-2)
    pos[..., 2] = data[..., 3]
    return q, pos


def joints_from_rel_traj(d):
    """recover_mode='joint_rel_traj' (motion_representation.py:349-371) -> [B, T, 22, 3]."""
    q, r_pos = recover_root_rot_pos_rel(torch.cat([d['root_rot_angle_vel'], d['root_l_vel'], d['root_height']], -1))
    positions = d['local_positions'][..., 3:]
    positions = positions.reshape(positions.shape[:-1] + (-1, 3))
    positions = qrot(qinv(q[..., None, :]).expand(positions.shape[:-1] + (4,)), positions)
    positions = positions.clone()
    positions[..., 0] += r_pos[..., 0:1]
    positions[..., 1] += r_pos[..., 1:2]
    return torch.cat([r_pos.unsqueeze(-2), positions], dim=-2)


def joints_from_smplx(d, body_model, return_verts=False, through_axis_angle=True):
    """recover_mode='smplx_params' (motion_representation.py:373-398) -> [B, T, 22, 3] (+ verts).
    `through_axis_angle=False` skips the R -> quaternion -> axis-angle -> Rodrigues round trip (what the HIP
    kernels do); the two must agree to rounding, which tests/test_geometry_oracle.py checks."""
    bs = len(d['smplx_rot_6d'])
    g_mat = rot6d_to_rotmat(d['smplx_rot_6d'].reshape(-1, 6))
    b_mat = rot6d_to_rotmat(d['smplx_body_pose_6d'].reshape(-1, 6))
    if through_axis_angle:
        g_aa = rotation_matrix_to_angle_axis(g_mat)
        b_aa = rotation_matrix_to_angle_axis(b_mat).reshape(-1, 63)
        out = body_model(betas=d['smplx_betas'].reshape(-1, 10), global_orient=g_aa, body_pose=b_aa,
                         transl=d['smplx_trans'].reshape(-1, 3), return_verts=return_verts)
        joints = out.joints[:, 0:22].reshape(bs, -1, 22, 3)
        return (joints, out.vertices.reshape(bs, joints.shape[1], -1, 3)) if return_verts else joints
    # direct FK on the Gram-Schmidt matrices
    N = g_mat.shape[0]
    R = torch.cat([g_mat[:, None], b_mat.reshape(N, 21, 3, 3)], dim=1)
    betas = d['smplx_betas'].reshape(-1, 10)
    Jr = body_model.J_regressor[:22]
    J = (Jr @ body_model.v_template)[None] + torch.einsum('jv,vck,nk->njc', Jr, body_model.shapedirs[:, :, :10], betas)
    G, P = [R[:, 0]], [J[:, 0]]
    for j in range(1, 22):
        p = int(body_model.parents[j])
        G.append(G[p] @ R[:, j])
        P.append(P[p] + (G[p] @ (J[:, j] - J[:, p])[..., None])[..., 0])
    joints = torch.stack(P, dim=1) + d['smplx_trans'].reshape(-1, 1, 3)
    return joints.reshape(bs, -1, 22, 3)


def perspective_projection(points, focal_length, camera_center):
    """K (p / p_z) with identity rotation (utils/other_utils.py:150-185)."""
    B = points.shape[0]
    K = torch.zeros([B, 3, 3], dtype=points.dtype)
    K[:, 0, 0], K[:, 1, 1], K[:, 2, 2] = focal_length[:, 0], focal_length[:, 1], 1.
    K[:, :-1, -1] = camera_center
    proj = points / points[:, :, -1].unsqueeze(-1)
    return torch.einsum('bij,bkj->bki', K, proj)[:, :, :-1]


# ------------------------------------------------------------------------------ guidance
def _denorm(x, mean, std):
    full = x[:, :, 0].permute(0, 2, 1)                                     # [B, T, 294]
    return full * std + mean


def _skating_term(joints, contact, fps=30, thr=0.1):
    v = (joints[:, 1:, FOOT_JOINTS] - joints[:, :-1, FOOT_JOINTS]) * fps
    v = torch.norm(v, dim=-1)
    mask = (v - thr).gt(0) * contact[:, :-1]
    return (v * mask).sum(), mask.sum()


def guide_skating(x0, mean, std, body_model, traj_feat_dim=22, through_axis_angle=True):
    """model/posenet.py:196-257 with compute_grad='x_0': returns grad [B, 294, 1, T] or None (= 0-d zero)."""
    with torch.enable_grad():
        x = x0.detach().clone().requires_grad_()
        full = _denorm(x, mean, std)
        d = split_repr(full)
        j_abs = joints_from_abs_traj(d)
        j_smpl = joints_from_smplx(d, body_model, through_axis_angle=through_axis_angle)
        contact = full[:, :, -4:].detach().clone()
        contact = (contact > 0.5).to(full.dtype)
        s_abs, n_abs = _skating_term(j_abs, contact)
        s_smpl, n_smpl = _skating_term(j_smpl, contact)
        if n_abs == 0 and n_smpl == 0:
            return None
        loss = 0.
        if n_smpl != 0:
            loss = loss + s_smpl / n_smpl
        if n_abs != 0:
            loss = loss + s_abs / n_abs
        g = torch.autograd.grad([-loss], [x])[0]
    g[:, 0:traj_feat_dim] = 0
    g[:, -4:] = 0
    return g


def guide_2d_projection(x0, mean, std, body_model, transf_matrix, focal_length, camera_center, keypoints_2d,
                        cam_R, cam_t, traj_feat_dim=22, through_axis_angle=True):
    """model/posenet.py:260-317 with compute_grad='x_0'."""
    with torch.enable_grad():
        x = x0.detach().clone().requires_grad_()
        d = split_repr(_denorm(x, mean, std))
        j = joints_from_smplx(d, body_model, through_axis_angle=through_axis_angle)          # [B, T, 22, 3]
        B, T = j.shape[:2]
        c2s = torch.linalg.inv(transf_matrix)
        R = c2s[:, 0:3, 0:3].unsqueeze(1).repeat(1, T, 1, 1).reshape(-1, 3, 3)
        t = c2s[:, 0:3, -1].unsqueeze(1).unsqueeze(1).repeat(1, T, 22, 1).reshape(-1, 22, 3)
        scene = torch.matmul(R, j.reshape(B * T, -1, 3).permute(0, 2, 1)).permute(0, 2, 1) + t
        cam = torch.matmul(torch.linalg.inv(cam_R), (scene - cam_t).permute(0, 2, 1)).permute(0, 2, 1)
        f = focal_length.unsqueeze(1).repeat(1, T, 1).reshape(-1, 2)
        c = camera_center.unsqueeze(1).repeat(1, T, 1).reshape(-1, 2)
        p2d = perspective_projection(cam, f, c).reshape(B, T, -1, 2)
        loss = (p2d - keypoints_2d[:, :T, :, 0:2]).abs() * keypoints_2d[:, :T, :, [-1]]
        loss = loss[:, :, PROJ_JOINTS].mean()
        g = torch.autograd.grad([-loss], [x])[0]
    g[:, 0:traj_feat_dim] = 0
    g[:, -4:] = 0
    return g
