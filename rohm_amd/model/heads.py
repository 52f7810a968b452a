"""Parameter containers with the reference's state_dict key names (model/heads.py).

None of these modules computes anything: the arithmetic of the hot path lives in
librohm_hip.so.  They exist so released checkpoints load with `strict=True` and so
`.to(device)`, `.parameters()`, `.eval()` behave as in the reference.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..utils.synth import sinusoid_table


def zero_module(module):
    """Zero-init helper used for the TrajControl 1x1 convs (model/heads.py:12-18)."""
    with torch.no_grad():
        for p in module.parameters():
            p.zero_()
    return module


class PositionalEncoding(nn.Module):
    """Holds the `pe` buffer [max_len, 1, d] (model/heads.py:112-129)."""

    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.register_buffer('pe', sinusoid_table(d_model, max_len))


class TimestepEmbedder(nn.Module):
    """Keys `time_embed.{0,2}.*` + shared `sequence_pos_encoder.pe` (model/heads.py:132-146)."""

    def __init__(self, latent_dim, sequence_pos_encoder):
        super().__init__()
        self.latent_dim = latent_dim
        self.sequence_pos_encoder = sequence_pos_encoder
        self.time_embed = nn.Sequential(nn.Linear(latent_dim, latent_dim), nn.SiLU(),
                                        nn.Linear(latent_dim, latent_dim))


class InputProcess(nn.Module):
    """Key `poseEmbedding.*`: Linear(body_feat_dim -> latent) (model/heads.py:149-160)."""

    def __init__(self, input_feats, latent_dim):
        super().__init__()
        self.input_feats, self.latent_dim = input_feats, latent_dim
        self.poseEmbedding = nn.Linear(input_feats, latent_dim)


class OutputProcess(nn.Module):
    """Key `poseFinal.*`: Linear(latent -> pose_feat_dim) (model/heads.py:163-176)."""

    def __init__(self, output_feats, latent_dim, nfeats):
        super().__init__()
        self.output_feats, self.latent_dim, self.nfeats = output_feats, latent_dim, nfeats
        self.poseFinal = nn.Linear(latent_dim, output_feats)


class Conv1dBlock(nn.Module):
    """Keys `block.0.*` (Conv1d) and `block.2.*` (GroupNorm(8)) (model/heads.py:90-106)."""

    def __init__(self, inp_channels, out_channels, kernel_size, n_groups=8):
        super().__init__()
        self.block = nn.ModuleList([
            nn.Conv1d(inp_channels, out_channels, kernel_size, padding=kernel_size // 2),
            nn.Identity(),
            nn.GroupNorm(n_groups, out_channels),
            nn.Identity(),
            nn.Identity(),
        ])


class ResidualTemporalBlock(nn.Module):
    """Keys `blocks.{0,1}.block.{0,2}.*`, `time_mlp.1.*`, `residual_conv.*` (model/heads.py:20-54)."""

    def __init__(self, inp_channels=4, out_channels=64, input_t=False, t_embed_dim=32, kernel_size=5):
        super().__init__()
        self.blocks = nn.ModuleList([Conv1dBlock(inp_channels, out_channels, kernel_size),
                                     Conv1dBlock(out_channels, out_channels, kernel_size)])
        self.input_t = input_t
        if input_t:
            self.time_mlp = nn.ModuleList([nn.Identity(), nn.Linear(t_embed_dim, out_channels), nn.Identity()])
        self.residual_conv = nn.Conv1d(inp_channels, out_channels, 1) if inp_channels != out_channels \
            else nn.Identity()


class Downsample1d(nn.Module):
    """Key `conv.*`: Conv1d(dim, dim, 3, stride 2, pad 1) (model/heads.py:72-78)."""

    def __init__(self, dim):
        super().__init__()
        self.conv = nn.Conv1d(dim, dim, 3, 2, 1)


class Upsample1d(nn.Module):
    """Key `conv.*`: ConvTranspose1d(dim, dim, 4, stride 2, pad 1) (model/heads.py:81-87)."""

    def __init__(self, dim):
        super().__init__()
        self.conv = nn.ConvTranspose1d(dim, dim, 4, 2, 1)
