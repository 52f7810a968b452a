"""Dataset-side SMPL-X work of the reference's loaders, batched on the device.

`data_loaders/dataloader_video.py:116-142` (PROX) and `:274-325` (EgoBody) call the body model ONCE PER FRAME while
they read a recording (plus a cam2world transform and `update_globalRT_for_smplx` per frame);
`data_loaders/dataloader_amass.py:194-206` calls it once per clip on the noise-perturbed parameters.  The loaders
themselves (file formats, pickles, OpenPose json) are out of scope; these two functions take the arrays the loaders
hold at those lines and return what those lines produce, for all frames in one launch."""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from .._lib import check, lib, ptr, stream_ptr
from ..body_model import native_for


def _dev32(x, device):
    return torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x).to(device=device, dtype=torch.float32).contiguous()


def frames_to_world(smplx_model, params, cam2world, device=None):
    """dataloader_video.py:121-142 for all N frames of a recording at once.

    params: dict with 'transl' [N,3], 'global_orient' [N,3], 'betas' [N,10], 'body_pose' [N,63] (numpy or tensors, the
    per-frame fitting results); cam2world [4,4].  Returns (joints_world [N,22,3] float32 tensor, smplx_world [N,79]
    float64 tensor = [global_orient, transl (both re-expressed in the world frame by update_globalRT_for_smplx,
    utils/other_utils.py:189-240), betas, body_pose] -- the rows the loader appends to `joints_list_world` /
    `smplx_list_world`)."""
    if device is None:
        device = smplx_model.v_template.device
    device = torch.device(device)
    nat = native_for(smplx_model, device)
    go, bp = _dev32(params['global_orient'], device), _dev32(params['body_pose'], device)
    be, tr = _dev32(params['betas'], device), _dev32(params['transl'], device)
    N = tr.shape[0]
    if go.shape != (N, 3) or bp.shape != (N, 63) or be.shape != (N, 10) or tr.shape != (N, 3):
        raise ValueError(f'expected [N,3] / [N,63] / [N,10] / [N,3], got {tuple(go.shape)} {tuple(bp.shape)} '
                         f'{tuple(be.shape)} {tuple(tr.shape)}')
    rigid = _dev32(cam2world, device)
    if rigid.shape != (4, 4):
        raise ValueError('cam2world must be [4, 4]')
    joints = torch.empty(N, 22, 3, device=device, dtype=torch.float32)
    ot = torch.empty(N, 6, device=device, dtype=torch.float64)
    check(lib().rohm_smplx_frames_to_world(nat.handle, ptr(go), ptr(bp), ptr(be), ptr(tr), ptr(rigid), N, ptr(joints),
                                           ptr(ot), stream_ptr(device)), 'rohm_smplx_frames_to_world')
    world = torch.cat([ot, be.double(), bp.double()], dim=-1)
    return joints, world


def noisy_clip_joints(smplx_model, params, device=None):
    """dataloader_amass.py:194-206: joints 0..21 [T, 22, 3] of a clip's (noise-perturbed) canonical SMPL-X parameters
    ('transl', 'global_orient', 'betas', 'body_pose' [T, 21, 3] or [T, 63]); face / hand poses are zero there."""
    if device is None:
        device = smplx_model.v_template.device
    device = torch.device(device)
    nat = native_for(smplx_model, device)
    go, bp = _dev32(params['global_orient'], device), _dev32(params['body_pose'], device)
    be, tr = _dev32(params['betas'], device), _dev32(params['transl'], device)
    N = tr.shape[0]
    pose = torch.cat([go.reshape(N, 1, 3), bp.reshape(N, 21, 3)], dim=1).contiguous()
    out = torch.empty(N, 22, 3, device=device, dtype=torch.float32)
    check(lib().rohm_smplx_joints(nat.handle, ptr(pose), 22, ptr(be), ptr(tr), N, ptr(out), 22, stream_ptr(device)),
          'rohm_smplx_joints')
    return out
