"""SMPL-X body-model layer backed by librohm_hip.so.

Plays the role of `smplx.create(model_type='smplx', gender='neutral', flat_hand_mean=True, use_pca=False)`
(model/posenet.py:57-58, test_amass_full.py:190-191) for the hot path: called as
`smplx_model(**{transl, global_orient, body_pose, betas, jaw_pose, ...})` it returns an object with
`.joints` (motion_representation.py:379-396).  Only joints 0..21 are populated (the hot path reads
`joints[:, 0:22]`); `.vertices` needs full linear blend skinning, which is a "next" row
and raises here.  Buffers carry smplx's own names so a checkpoint's `smplx_model.*` tensors load.
"""
from __future__ import annotations

import ctypes as C
import types

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import check, lib, ptr, stream_ptr


class _NativeSMPLX:
    def __init__(self, layer, device):
        self.device = device
        f = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
        vt, sd, jr = f(layer.v_template), f(layer.shapedirs), f(layer.J_regressor)
        par = layer.parents.detach().to(torch.int32).cpu().contiguous()
        self.handle = C.c_void_p()
        torch.cuda.synchronize(device)
        with torch.cuda.device(device):
            check(lib().rohm_smplx_create(C.byref(self.handle), ptr(vt), ptr(sd), sd.shape[-1], ptr(jr),
                                          C.c_void_p(par.data_ptr()), vt.shape[0], jr.shape[0], device.index or 0),
                  'rohm_smplx_create')
        self._ws = None

    def workspace(self, B, T):
        n = lib().rohm_guidance_workspace_bytes(B, T)
        if self._ws is None or self._ws.numel() < n:
            self._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
        return self._ws

    def __del__(self):
        try:
            if self.handle:
                lib().rohm_smplx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def native_for(body_model, device):
    """Native handle for any body model exposing v_template / shapedirs / J_regressor / parents
    (our SMPLXLayer or a real `smplx.SMPLX`), cached on the module."""
    device = torch.device(device)
    if device.type != 'cuda':
        raise _lib.RohmHipError('the SMPL-X kernels run only on an AMD GPU (no CPU fallback)')
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    cache = body_model.__dict__.setdefault('_rohm_native', {})
    key = (str(device), body_model.v_template.data_ptr(), body_model.shapedirs.data_ptr())
    if key not in cache:
        cache.clear()
        cache[key] = _NativeSMPLX(body_model, device)
    return cache[key]


class SMPLXLayer(nn.Module):
    def __init__(self, v_template, shapedirs, J_regressor, parents, posedirs=None, lbs_weights=None, faces=None):
        super().__init__()
        self.register_buffer('v_template', torch.as_tensor(v_template, dtype=torch.float32))
        self.register_buffer('shapedirs', torch.as_tensor(shapedirs, dtype=torch.float32))
        self.register_buffer('J_regressor', torch.as_tensor(J_regressor, dtype=torch.float32))
        self.register_buffer('parents', torch.as_tensor(parents, dtype=torch.long))
        if posedirs is not None:
            self.register_buffer('posedirs', torch.as_tensor(posedirs, dtype=torch.float32))
        if lbs_weights is not None:
            self.register_buffer('lbs_weights', torch.as_tensor(lbs_weights, dtype=torch.float32))
        self.faces = faces

    @classmethod
    def from_tensors(cls, t):
        return cls(t['v_template'], t['shapedirs'], t['J_regressor'], t['parents'], t.get('posedirs'),
                   t.get('lbs_weights'))

    @classmethod
    def from_npz(cls, path, num_betas=10, num_expression_coeffs=10):
        """Load an SMPLX_*.npz model file (keys v_template, shapedirs, posedirs, J_regressor, kintree_table,
        weights, f) the way smplx does: first `num_betas` shape + first expression components."""
        d = np.load(path, allow_pickle=True)
        sd = np.asarray(d['shapedirs'], dtype=np.float32)
        sd = np.concatenate([sd[:, :, :num_betas], sd[:, :, 300:300 + num_expression_coeffs]], axis=2) \
            if sd.shape[2] >= 300 + num_expression_coeffs else sd
        parents = np.asarray(d['kintree_table'])[0].astype(np.int64)
        parents[0] = -1
        pd = np.asarray(d['posedirs'], dtype=np.float32)
        pd = pd.reshape(-1, pd.shape[-1]).T
        return cls(d['v_template'], sd, d['J_regressor'], parents, pd, d['weights'], faces=d['f'])

    def forward(self, betas=None, global_orient=None, body_pose=None, transl=None, return_verts=False, **unused):
        """Axis-angle in, `.joints` [N, 127, 3] out (rows 22.. are zero: not produced by the hot path)."""
        _lib.require_hip(betas, global_orient, body_pose, transl)
        nat = native_for(self, betas.device)
        N = betas.shape[0]
        pose = torch.cat([global_orient.reshape(N, 1, 3), body_pose.reshape(N, -1, 3)], dim=1).float().contiguous()
        j22 = torch.empty(N, 22, 3, device=betas.device, dtype=torch.float32)
        check(lib().rohm_smplx_joints(nat.handle, ptr(pose), pose.shape[1], ptr(betas.float().contiguous()),
                                      ptr(transl.float().contiguous()), N, ptr(j22), 22, stream_ptr(betas.device)),
              'rohm_smplx_joints')
        joints = torch.zeros(N, 127, 3, device=betas.device, dtype=torch.float32)
        joints[:, :22] = j22
        out = types.SimpleNamespace(joints=joints)
        if return_verts:
            raise NotImplementedError('vertices need full LBS (SURVEY.md §8f N1/N3): not part of the hot path')
        return out
