"""Shared test helpers (seeded inputs identical to oracle/make_golden.py)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def seeded(seed, *shape):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(g.standard_normal(size=shape).astype(np.float32))


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def cpu_noise_sequence(torch_seed, shape, steps, trajnet_layout=False):
    """The noise the reference draws on CPU: one randn(*shape), then one randn_like per step
    (gaussian_diffusion_posenet.py:613,458).

    `trajnet_layout`: TrajNet returns a permuted *view* ('b t h -> b h t', trajnet.py:274), so from the
    second step on `x` is dense but channel-major in memory and `randn_like(x)` (preserve_format) fills it
    in that memory order: logically noise[b, t, c] = stream[b, c, t].  Reproduced here so the injected
    noise equals the reference's CPU stream element for element."""
    torch.manual_seed(torch_seed)
    x_T = torch.randn(*shape)
    out = []
    for k in range(steps):
        if trajnet_layout and k > 0:
            B, T, Cc = shape
            # same strides as the reference's x => same (non-vectorised, memory-order) normal_ path
            out.append(torch.randn_like(torch.empty(B, Cc, T).permute(0, 2, 1)))
        else:
            out.append(torch.randn(*shape))
    return x_T, out


def cpu_noise_stream(torch_seed, plan):
    """The draws of a whole free-running scheme from ONE seeded global generator, stage after stage, as the reference makes
    them: `plan` = [(kind, shape, steps), ...] with kind 'traj' (TrajNet's permuted-view randn_like, see
    cpu_noise_sequence) or 'pose'.  Returns {'traj': [(x_T, [noise]) per stage], 'pose': [...]}."""
    torch.manual_seed(torch_seed)
    runs = {'traj': [], 'pose': []}
    for kind, shape, steps in plan:
        x_T = torch.randn(*shape)
        out = []
        for k in range(steps):
            if kind == 'traj' and k > 0:
                B, T, Cc = shape
                out.append(torch.randn_like(torch.empty(B, Cc, T).permute(0, 2, 1)))
            else:
                out.append(torch.randn(*shape))
        runs[kind].append((x_T, out))
    return runs


class PoseDataset:
    """Minimal stand-in for the attributes PoseNet reads from its dataset (posenet.py:207,210,289,296)."""
    pose_feat_dim = 272
    traj_feat_dim = 22
    body_feat_dim = 294
    joints_num = 22

    def __init__(self, mean=None, std=None):
        self.Mean = np.zeros(294, np.float32) if mean is None else mean
        self.Std = np.ones(294, np.float32) if std is None else std


def max_abs(a, b):
    return float((a.double() - b.double()).abs().max())
