"""TEST INFRASTRUCTURE ONLY: CPU restatement of the dataset-side per-frame SMPL-X work.

`data_loaders/dataloader_video.py:116-142` (PROX) and `:274-300` (EgoBody) run, for EVERY frame of a recording: one
SMPL-X forward -> joints in camera coordinates -> joints to world (cam2world) -> `update_globalRT_for_smplx`
(utils/other_utils.py:189-240: global orientation / translation re-expressed in the world frame, float64 numpy +
scipy Rotation).  `data_loaders/dataloader_amass.py:194-206` runs one batched SMPL-X forward on the noise-perturbed
parameters of a clip.  Restated here batched over the frames; pinned to the reference's own `update_globalRT_for_smplx`
(tests/golden/frames.npz, oracle/make_golden.py::golden_frames)."""
from __future__ import annotations

import numpy as np
import torch
from scipy.spatial.transform import Rotation as R


def update_global_rt(global_orient, transl, delta_T, rigid):
    """utils/other_utils.py:221-240 for a batch of frames (float64, like the reference)."""
    bs = len(transl)
    body_mat = np.zeros([bs, 4, 4])
    body_mat[:, :-1, :-1] = R.from_rotvec(global_orient).as_matrix()
    body_mat[:, :-1, -1] = transl + delta_T
    body_mat[:, -1, -1] = 1
    new = np.matmul(np.repeat(np.expand_dims(rigid, 0), bs, axis=0), body_mat)
    return R.from_matrix(new[:, :-1, :-1]).as_rotvec().reshape(-1, 3), (new[:, :-1, -1] - delta_T).reshape(-1, 3)


def frames_to_world(body_model, params, cam2world, joints_num=22):
    """dataloader_video.py:121-142 for all frames at once.  params: dict of float32 numpy arrays transl [N,3],
    global_orient [N,3], betas [N,10], body_pose [N,63]; cam2world [4,4] float32.
    Returns joints_world [N, 22, 3] float32 and smplx_world [N, 79] float64."""
    tp = {k: torch.tensor(v) for k, v in params.items()}
    joints_cam = body_model(return_verts=True, **tp).joints[:, 0:joints_num, :]                       # :127-128
    c2w = torch.from_numpy(cam2world).float()
    cam_R, cam_t = c2w[:3, :3].reshape(3, 3), c2w[:3, 3].reshape(1, 3)
    joints = torch.matmul(cam_R, joints_cam.permute(0, 2, 1)).permute(0, 2, 1) + cam_t              # :131
    delta_T = joints_cam[:, 0].detach().cpu().numpy() - params['transl']                             # :140
    go, tr = update_global_rt(params['global_orient'], params['transl'], delta_T, cam2world)
    world = np.concatenate([go, tr, params['betas'], params['body_pose']], axis=-1)                   # :141-142
    return joints.detach().numpy(), world


def noisy_clip_joints(body_model, params):
    """dataloader_amass.py:194-206: joints 0..21 of the noise-perturbed canonical parameters of a clip."""
    tp = {k: torch.FloatTensor(np.asarray(v, dtype=np.float32)) for k, v in params.items()}
    n = tp['transl'].shape[0]
    z = lambda d: torch.zeros(n, d)
    return body_model(jaw_pose=z(3), leye_pose=z(3), reye_pose=z(3), left_hand_pose=z(45), right_hand_pose=z(45),
                      expression=z(10), **tp).joints[:, 0:22].detach().cpu().numpy()
