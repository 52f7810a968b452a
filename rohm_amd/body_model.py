"""SMPL-X body-model layer backed by librohm_hip.so.

Plays the role of `smplx.create(model_type='smplx', gender='neutral', flat_hand_mean=True, use_pca=False)`
(model/posenet.py:57-58, test_amass_full.py:190-191) for the hot path: called as
`smplx_model(**{transl, global_orient, body_pose, betas, jaw_pose, ...})` it returns an object with
`.joints` (motion_representation.py:379-396).  By default only joints 0..21 are populated (the hot path reads
`joints[:, 0:22]`, joints-only FK); `return_verts=True` runs full linear blend skinning (`rohm_smplx_forward`,
csrc/lbs.hip) and also returns `.vertices`.  Buffers carry smplx's own names so a checkpoint's `smplx_model.*`
tensors load.
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
            self.has_lbs = False
            pd, lw = getattr(layer, 'posedirs', None), getattr(layer, 'lbs_weights', None)
            if pd is not None and lw is not None:
                pd, lw = f(pd), f(lw)
                check(lib().rohm_smplx_set_skinning(self.handle, ptr(vt), ptr(sd), sd.shape[-1], ptr(pd), pd.shape[0],
                                                    ptr(lw)), 'rohm_smplx_set_skinning')
                self.has_lbs = True
            self.num_verts, self.num_joints = vt.shape[0], jr.shape[0]
        self._ws = None
        self._lbs_ws = None

    def lbs_workspace(self, N):
        n = lib().rohm_smplx_lbs_workspace_bytes(self.handle, N)
        if self._lbs_ws is None or self._lbs_ws.numel() < n:
            self._lbs_ws = None
            self._lbs_ws = torch.empty(n, dtype=torch.uint8, device=self.device)
        return self._lbs_ws

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


def lbs_forward(nat, pose, pose_kind, betas, transl, want_verts=True):
    """rohm_smplx_forward over any number of frames: pose [N, n_pose, 3 | 6] (pose_kind 0 axis-angle / 1 six-D),
    returns (joints [N, J, 3], verts [N, V, 3] or None)."""
    if not nat.has_lbs:
        raise _lib.RohmHipError('vertices need posedirs and lbs_weights on the body-model layer')
    dev, N, J = pose.device, pose.shape[0], nat.num_joints
    verts = torch.empty(N, nat.num_verts, 3, device=dev, dtype=torch.float32) if want_verts else None
    jall = torch.empty(N, J, 3, device=dev, dtype=torch.float32)
    step = 16384                                              # frames per launch (workspace 126 kB per frame)
    for s0 in range(0, N, step):
        n = min(step, N - s0)
        ws = nat.lbs_workspace(n)
        check(lib().rohm_smplx_forward(nat.handle, ptr(pose[s0:s0 + n]), pose.shape[1], pose_kind, ptr(betas[s0:s0 + n]),
                                       ptr(transl[s0:s0 + n]), n, ptr(jall[s0:s0 + n]), J,
                                       ptr(verts[s0:s0 + n]) if want_verts else None, ptr(ws), ws.numel(),
                                       stream_ptr(dev)), 'rohm_smplx_forward')
    return jall, verts


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

    def forward(self, betas=None, global_orient=None, body_pose=None, transl=None, return_verts=False, jaw_pose=None,
                leye_pose=None, reye_pose=None, left_hand_pose=None, right_hand_pose=None, expression=None, **unused):
        """Axis-angle in, object with `.joints` [N, 127, 3] (and `.vertices` [N, V, 3] with `return_verts=True`) out.
        Joints-only (default): rows 22.. are zero -- the hot path reads `joints[:, 0:22]`.  With vertices: full LBS,
        the 55 kinematic joints are filled (smplx's 72 extra landmark joints stay zero), face / hand poses are
        honoured when given, expression coefficients must be zero (as at every reference call site)."""
        _lib.require_hip(betas, global_orient, body_pose, transl)
        nat = native_for(self, betas.device)
        dev = betas.device
        N = betas.shape[0]
        parts = [global_orient.reshape(N, 1, 3), body_pose.reshape(N, -1, 3)]
        joints = torch.zeros(N, 127, 3, device=dev, dtype=torch.float32)
        b32, t32 = betas.float().contiguous(), transl.float().contiguous()
        if not return_verts:
            pose = torch.cat(parts, dim=1).float().contiguous()
            j22 = torch.empty(N, 22, 3, device=dev, dtype=torch.float32)
            check(lib().rohm_smplx_joints(nat.handle, ptr(pose), pose.shape[1], ptr(b32), ptr(t32), N, ptr(j22), 22,
                                          stream_ptr(dev)), 'rohm_smplx_joints')
            joints[:, :22] = j22
            return types.SimpleNamespace(joints=joints)
        if not nat.has_lbs:
            raise _lib.RohmHipError('return_verts=True needs posedirs and lbs_weights on the body-model layer')
        if expression is not None and float(expression.abs().max()) != 0.0:
            raise NotImplementedError('non-zero expression coefficients are not supported')
        extra = [jaw_pose, leye_pose, reye_pose, left_hand_pose, right_hand_pose]
        if any(e is not None for e in extra):
            sizes = [1, 1, 1, 15, 15]
            parts += [(e.reshape(N, -1, 3) if e is not None else torch.zeros(N, k, 3, device=dev)) for e, k in zip(extra, sizes)]
        pose = torch.cat(parts, dim=1).float().contiguous()
        J = nat.num_joints
        if pose.shape[1] > J:
            raise ValueError(f'{pose.shape[1]} joint rotations for a {J}-joint model')
        jall, verts = lbs_forward(nat, pose, 0, b32, t32)
        joints[:, :J] = jall
        return types.SimpleNamespace(joints=joints, vertices=verts)
