"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the between-stage trajectory
re-derivation of the reference drivers.

What the reference does between TrajNet and PoseNet (test_amass_full.py:262-311, test_prox_egobody.py:238-287):
de-normalise the trajectory-side representation, run SMPL-X on every frame
(`recover_from_repr_smpl(..., 'smplx_params', return_verts=True)`, vertices unused), copy joints and
parameters to the host and, one sequence at a time, re-compute the motion representation from them with
`get_repr_smplx` (data_loaders/motion_representation.py:187-282), re-normalise with the pose dataset's
statistics and keep the first 22 (trajectory) channels.

Dtype flow follows the reference exactly (it decides the last bits): joint positions and the quaternion
helpers are float32 (`qbetween_np`/`qmul_np`/`qrot_np` cast to float32, quaternion.py:21-23,126-135,397-406);
the forward direction, the scipy rotation matrices, the angular velocity and the final normalisation are
float64; the result is rounded to float32 when it is written into `cond` (test_amass_full.py:336).

Pinned against the reference's own `get_repr_smplx` and driver lines by oracle/make_golden.py ->
tests/golden/rederive.npz (bit-exact on the golden inputs).  The SMPL-X joints feeding it come from
oracle/geometry.py::BodyModel ("parity unpinned", third-party smplx==0.1.28 absent).
"""
import numpy as np
import torch
from scipy.spatial.transform import Rotation

from . import geometry as G

FID_L, FID_R = [7, 10], [8, 11]              # motion_representation.py:12
# motion_representation.py:201 unpacks face_joint_indx = [2, 1, 17, 16] as `l_hip, r_hip, sdr_r, sdr_l`, i.e. the names
# used inside get_repr_smplx are l_hip = 2, r_hip = 1 (swapped against the module-level r_hip, l_hip = 2, 1 of :16).
L_HIP, R_HIP, SDR_R, SDR_L = 2, 1, 17, 16


def foot_detect(positions, thres):
    """motion_representation.py:23-44 with up_axis='z': contact = slow AND low.  positions [T,22,3] float32."""
    vel_thr = np.array([thres, thres])
    h_thr = np.array([0.18, 0.15])
    out = []
    for fid in (FID_L, FID_R):
        d = positions[1:, fid] - positions[:-1, fid]
        sq = d[..., 0] ** 2 + d[..., 1] ** 2 + d[..., 2] ** 2
        h = positions[:-1, fid, 2]
        out.append(((sq < vel_thr) & (h < h_thr)).astype(float))
    return out[0], out[1]


def _f32(a):
    return torch.from_numpy(np.ascontiguousarray(a)).float()


def _qnormalize(q):
    return q / torch.norm(q, dim=-1, keepdim=True)            # quaternion.py:26-28


def _qbetween(v0, v1):
    """quaternion.py:385-394 in float32."""
    v0, v1 = _f32(v0), _f32(v1)
    v = torch.cross(v0, v1, dim=-1)
    w = torch.sqrt((v0 ** 2).sum(-1, keepdim=True) * (v1 ** 2).sum(-1, keepdim=True)) + (v0 * v1).sum(-1, keepdim=True)
    return _qnormalize(torch.cat([w, v], dim=-1)).numpy()


def _qinv(q):
    m = np.array([1., -1., -1., -1.], np.float32)
    return (_f32(q) * torch.from_numpy(m)).numpy()


def _qmul(q, r):
    """quaternion.py:31-49: outer product terms[i][j] = r_i q_j, fixed summation order, float32."""
    q, r = _f32(q).reshape(-1, 4), _f32(r).reshape(-1, 4)
    t = torch.bmm(r.view(-1, 4, 1), q.view(-1, 1, 4))
    w = t[:, 0, 0] - t[:, 1, 1] - t[:, 2, 2] - t[:, 3, 3]
    x = t[:, 0, 1] + t[:, 1, 0] - t[:, 2, 3] + t[:, 3, 2]
    y = t[:, 0, 2] + t[:, 1, 3] + t[:, 2, 0] - t[:, 3, 1]
    z = t[:, 0, 3] - t[:, 1, 2] + t[:, 2, 1] + t[:, 3, 0]
    return torch.stack((w, x, y, z), dim=1).numpy()


def _qrot(q, v):
    shape = v.shape
    return G.qrot(_f32(q).reshape(-1, 4), _f32(v).reshape(-1, 3)).reshape(shape).numpy()


def angular_velocity(rot, d_rot):
    """utils/other_utils.py:264-277: vee of (dR R^T), symmetric entries averaged."""
    w = np.matmul(d_rot, np.transpose(rot, (0, 2, 1)))
    return np.stack([(-w[:, 1, 2] + w[:, 2, 1]) / 2.0, (w[:, 0, 2] - w[:, 2, 0]) / 2.0,
                     (-w[:, 0, 1] + w[:, 1, 0]) / 2.0], axis=-1)


def get_repr_smplx(positions, params, feet_vel_thre=5e-5):
    """motion_representation.py:187-282.  positions [T,22,3] float32 numpy, params: transl [T,3],
    global_orient [T,3] (axis-angle), body_pose [T,63], betas [T,10].  Returns the dict of T-1 frame features."""
    T = positions.shape[0]
    feet_l, feet_r = foot_detect(positions, feet_vel_thre)

    across = (positions[:, R_HIP] - positions[:, L_HIP]) + (positions[:, SDR_R] - positions[:, SDR_L])
    across = across / np.sqrt((across ** 2).sum(axis=-1))[:, None]
    forward = np.cross(np.array([[0, 0, 1]]), across, axis=-1)                    # float64 from here
    forward = forward / np.sqrt((forward ** 2).sum(axis=-1))[..., None]

    quat = _qbetween(forward, np.array([[0, 1, 0]]).repeat(T, axis=0))            # float32
    if np.isnan(quat).sum() > 0:                                                  # only the FIRST NaN frame is patched
        k = np.where(np.isnan(quat))[0][0]
        quat[k] = quat[k - 1]
    quat[0] = np.array([1.0, 0.0, 0.0, 0.0])
    quat_vel = _qmul(quat[1:], _qinv(quat[:-1]))

    root = positions[:, 0]
    root_l_vel = _qrot(quat[1:], (positions[1:, 0] - positions[:-1, 0]).copy())   # rotated by the NEXT frame's q
    ang = np.arctan2(quat[:, 3:4], quat[:, 0:1])
    ang_vel = np.arctan2(quat_vel[:, 3:4], quat_vel[:, 0:1])

    local = positions.copy()
    local[..., 0] -= local[:, 0:1, 0]
    local[..., 1] -= local[:, 0:1, 1]
    local = _qrot(np.repeat(quat[:, None], 22, axis=1), local)
    local_vel = _qrot(np.repeat(quat[:-1, None], 22, axis=1), positions[1:] - positions[:-1])

    rot = Rotation.from_rotvec(params['global_orient']).as_matrix()               # float64
    rot6d = rot[..., :-1].reshape(-1, 6)
    rot_vel = angular_velocity(rot[:-1], rot[1:] - rot[:-1])
    trans = params['transl']
    body = Rotation.from_rotvec(params['body_pose'].reshape(-1, 3)).as_matrix().reshape(T, -1, 3, 3)
    body6d = body[..., :-1].reshape(T, -1, 6)
    n = T - 1
    return {'root_rot_angle': ang[:-1], 'root_rot_angle_vel': ang_vel, 'root_l_pos': root[:-1, [0, 1]],
            'root_l_vel': root_l_vel[:, [0, 1]], 'root_height': root[:-1, 2:3], 'smplx_rot_6d': rot6d[:-1],
            'smplx_rot_vel': rot_vel, 'smplx_trans': trans[:-1], 'smplx_trans_vel': (trans[1:] - trans[:-1]).copy(),
            'local_positions': local[:-1].reshape(n, -1), 'local_vel': local_vel.reshape(n, -1),
            'smplx_body_pose_6d': body6d[:-1].reshape(n, -1), 'smplx_betas': params['betas'][:-1],
            'foot_contact': np.concatenate([feet_l, feet_r], axis=-1)}


def facing_margin(positions):
    """Conditioning of the facing-direction channels (0, 1, 4, 5) per frame, as two positive numbers whose
    smallness marks frames where the reference's own float32 result is ill-conditioned (tests widen their
    tolerance on exactly those frames): (|across_xy| in metres, w = 1 + forward.y).  The facing angle moves by
    ~d/|across_xy| for a joint displacement d (hips/shoulders stacked vertically -> no facing direction), and
    qbetween's w = |f| + f.y cancels when the body faces -y."""
    ac = (positions[..., R_HIP, :] - positions[..., L_HIP, :]) + (positions[..., SDR_R, :] - positions[..., SDR_L, :])
    ac = ac.astype(np.float64)
    raw_xy = np.sqrt(ac[..., 0] ** 2 + ac[..., 1] ** 2)
    with np.errstate(invalid='ignore', divide='ignore'):
        fy = ac[..., 0] / raw_xy                     # forward = (-a_y, a_x, 0) / |a_xy|
    return raw_xy, 1.0 + fy


def full_repr(d):
    return np.concatenate([d[k] for k in G.REPR_LIST], axis=-1)


def rederive_traj(repr_norm, mean_in, std_in, mean_out, std_out, body_model, n_keep=22, return_full=False,
                  return_joints=False):
    """test_amass_full.py:262-311.  repr_norm [B,T,294] float32 tensor normalised with (mean_in, std_in) (the
    trajectory dataset); returns the first `n_keep` channels of the re-derived representation normalised with
    (mean_out, std_out) (the pose dataset): float64 numpy [B,T-1,n_keep], as the reference holds it before the
    assignment into the float32 `cond`."""
    x = repr_norm.detach().cpu().numpy() * std_in + mean_in          # float32 numpy
    d = G.split_repr(torch.from_numpy(x))
    joints = G.joints_from_smplx(d, body_model).detach().cpu().numpy()
    out = []
    for b in range(x.shape[0]):
        g_aa = G.rotation_matrix_to_angle_axis(G.rot6d_to_rotmat(d['smplx_rot_6d'][b]))
        b_aa = G.rotation_matrix_to_angle_axis(G.rot6d_to_rotmat(d['smplx_body_pose_6d'][b].reshape(-1, 6)))
        params = {'transl': d['smplx_trans'][b].numpy(), 'global_orient': g_aa.numpy(),
                  'body_pose': b_aa.reshape(-1, 63).numpy(), 'betas': d['smplx_betas'][b].numpy()}
        full = full_repr(get_repr_smplx(joints[b], params))
        full = (full - mean_out) / std_out
        out.append(full if return_full else full[:, :n_keep])
    return (np.asarray(out), joints) if return_joints else np.asarray(out)
