"""Functional restatement of the reference networks on CPU (test oracle).

All functions take a flat ``state_dict`` (name -> tensor) with the reference's key
names and compute in ``dtype`` (fp32 default; fp64 gives a tighter yardstick).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _cast(sd, dtype):
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}


# --------------------------------------------------------------------------- PoseNet
def layer_norm(x, w, b, eps=1e-5):
    """nn.LayerNorm over the last dim, biased variance (torch default; posenet.py:63-69)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_erf(x):
    """Exact-erf GELU (activation="gelu", posenet.py:67)."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def timestep_token(sd, t):
    """`TimestepEmbedder.forward`, model/heads.py:145-146: pe[t] -> Linear -> SiLU -> Linear."""
    pe = sd['sequence_pos_encoder.pe'][:, 0]                       # [5000, D]
    e = pe[t]                                                      # [B, D]
    e = e @ sd['embed_timestep.time_embed.0.weight'].T + sd['embed_timestep.time_embed.0.bias']
    e = e * torch.sigmoid(e)
    e = e @ sd['embed_timestep.time_embed.2.weight'].T + sd['embed_timestep.time_embed.2.bias']
    return e


def encoder_layer(sd, p, x, n_head):
    """One post-norm `nn.TransformerEncoderLayer` (posenet.py:63-69); x is [B, S, D]."""
    B, S, D = x.shape
    dh = D // n_head
    qkv = x @ sd[p + 'self_attn.in_proj_weight'].T + sd[p + 'self_attn.in_proj_bias']
    q, k, v = qkv.split(D, dim=-1)
    q = q.view(B, S, n_head, dh).transpose(1, 2)                   # [B, H, S, dh]
    k = k.view(B, S, n_head, dh).transpose(1, 2)
    v = v.view(B, S, n_head, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    a = torch.softmax(s, dim=-1) @ v                               # [B, H, S, dh]
    a = a.transpose(1, 2).reshape(B, S, D)
    a = a @ sd[p + 'self_attn.out_proj.weight'].T + sd[p + 'self_attn.out_proj.bias']
    x = layer_norm(x + a, sd[p + 'norm1.weight'], sd[p + 'norm1.bias'])
    f = gelu_erf(x @ sd[p + 'linear1.weight'].T + sd[p + 'linear1.bias'])
    f = f @ sd[p + 'linear2.weight'].T + sd[p + 'linear2.bias']
    return layer_norm(x + f, sd[p + 'norm2.weight'], sd[p + 'norm2.bias'])


def posenet_forward(sd, x_t, cond, t, n_head=4, traj_feat_dim=22, dtype=torch.float32,
                    return_hidden=False):
    """`PoseNet.forward`, model/posenet.py:75-96.

    x_t, cond: [B, C, 1, T]; t: int64 [B] -> [B, C, 1, T] (channels < traj_feat_dim
    copied from cond, the remaining 272 predicted).
    """
    sd = _cast(sd, dtype)
    x_t, cond = x_t.to(dtype), cond.to(dtype)
    B, C, _, T = x_t.shape
    n_layer = 1 + max(int(k.split('.')[2]) for k in sd if k.startswith('seqTransEncoder.layers.'))
    emb = timestep_token(sd, t)                                    # [B, D]
    xs = x_t[:, :, 0].permute(0, 2, 1)                             # [B, T, C]  (heads.py:156-160)
    cs = cond[:, :, 0].permute(0, 2, 1)
    h = (xs @ sd['input_process.poseEmbedding.weight'].T + sd['input_process.poseEmbedding.bias']
         + cs @ sd['input_process_cond.poseEmbedding.weight'].T
         + sd['input_process_cond.poseEmbedding.bias'])
    seq = torch.cat([emb[:, None], h], dim=1)                      # token 0 = timestep (posenet.py:90)
    seq = seq + sd['sequence_pos_encoder.pe'][:T + 1, 0][None]     # (posenet.py:91)
    hidden = [seq]
    for i in range(n_layer):
        seq = encoder_layer(sd, f'seqTransEncoder.layers.{i}.', seq, n_head)
        hidden.append(seq)
    out = seq[:, 1:] @ sd['output_process.poseFinal.weight'].T + sd['output_process.poseFinal.bias']
    out = out.permute(0, 2, 1)[:, :, None]                         # [B, 272, 1, T] (heads.py:171-176)
    out = torch.cat([cond[:, :traj_feat_dim], out], dim=1)         # (posenet.py:94-95)
    return (out, hidden) if return_hidden else out


# --------------------------------------------------------------------------- TrajNet
def mish(x):
    return x * torch.tanh(F.softplus(x))


def conv_block(sd, p, x, n_groups=8):
    """`Conv1dBlock`, model/heads.py:90-106: Conv1d(k, pad k//2) -> GroupNorm(8) -> Mish."""
    w = sd[p + '0.weight']
    x = F.conv1d(x, w, sd[p + '0.bias'], padding=w.shape[-1] // 2)
    x = F.group_norm(x, n_groups, sd[p + '2.weight'], sd[p + '2.bias'], eps=1e-5)
    return mish(x)


def res_block(sd, p, x, temb):
    """`ResidualTemporalBlock.forward`, model/heads.py:43-54."""
    out = conv_block(sd, p + '.blocks.0.block.', x)
    if (p + '.time_mlp.1.weight') in sd:
        tb = mish(temb) @ sd[p + '.time_mlp.1.weight'].T + sd[p + '.time_mlp.1.bias']
        out = out + tb[:, :, None]
    out = conv_block(sd, p + '.blocks.1.block.', out)
    if (p + '.residual_conv.weight') in sd:
        res = F.conv1d(x, sd[p + '.residual_conv.weight'], sd[p + '.residual_conv.bias'])
    else:
        res = x
    return out + res


def down(sd, p, x):
    """`Downsample1d`, heads.py:72-78: Conv1d(k3, s2, p1)."""
    return F.conv1d(x, sd[p + '.conv.weight'], sd[p + '.conv.bias'], stride=2, padding=1)


def up(sd, p, x):
    """`Upsample1d`, heads.py:81-87: ConvTranspose1d(k4, s2, p1)."""
    return F.conv_transpose1d(x, sd[p + '.conv.weight'], sd[p + '.conv.bias'], stride=2, padding=1)


def time_embedding(sd, t, dim=32):
    """`SinusoidalPosEmb` + time_mlp, heads.py:57-69, trajnet.py:120-125."""
    half = dim // 2
    w = sd['time_mlp.1.weight']
    freq = torch.exp(torch.arange(half, dtype=w.dtype) * -(math.log(10000) / (half - 1)))
    e = t.to(w.dtype)[:, None] * freq[None]
    e = torch.cat([e.sin(), e.cos()], dim=-1)
    e = mish(e @ w.T + sd['time_mlp.1.bias'])
    return e @ sd['time_mlp.3.weight'].T + sd['time_mlp.3.bias']


def controlnet_forward(sd, control_cond, h_cond, temb):
    """`ControlNet.forward`, model/trajnet.py:43-75. control_cond: [B, T, 272]."""
    c = 'controlnet.'
    x = control_cond.permute(0, 2, 1)
    x = F.conv1d(x, sd[c + 'control_zero_conv_0.weight'], sd[c + 'control_zero_conv_0.bias'])
    outs = []
    for i in range(4):
        x = res_block(sd, c + f'control_enc{i + 1}', x, temb)
        outs.append(F.conv1d(x, sd[c + f'control_zero_conv_{i + 1}.weight'],
                             sd[c + f'control_zero_conv_{i + 1}.bias']))
        x = down(sd, c + f'control_downsample{i + 1}', torch.cat([x, h_cond[i]], dim=1))
    x = res_block(sd, c + 'control_mid_block1', x, temb)
    x = res_block(sd, c + 'control_mid_block2', x, temb)
    outs.append(F.conv1d(x, sd[c + 'control_zero_conv_mid.weight'], sd[c + 'control_zero_conv_mid.bias']))
    return outs


def trajnet_forward(sd, x_t, cond, t, control_cond=None, dtype=torch.float32):
    """`TrajNet.forward`, model/trajnet.py:177-275. x_t, cond: [B, T, 13] -> [B, T, 13]."""
    sd = _cast(sd, dtype)
    x_t, cond = x_t.to(dtype), cond.to(dtype)
    trajcontrol = any(k.startswith('controlnet.') for k in sd)
    temb = time_embedding(sd, t)                                   # [B, 32]
    c = cond.permute(0, 2, 1)
    h_cond = []
    for i in range(4):                                             # trajnet.py:192-208
        c = res_block(sd, f'cond_enc{i + 1}', c, None)
        h_cond.append(c)
        if i < 3:
            c = down(sd, f'cond_downsample{i + 1}', c)
    if trajcontrol:
        ctrl = controlnet_forward(sd, control_cond.to(dtype), h_cond, temb)
    x = x_t.permute(0, 2, 1)
    h_diff = []
    for i in range(4):                                             # trajnet.py:220-234
        x = res_block(sd, f'diff_enc{i + 1}', x, temb)
        h_diff.append(x)
        x = down(sd, f'diff_downsample{i + 1}', torch.cat([x, h_cond[i]], dim=1))
    x = res_block(sd, 'diff_mid_block1', x, temb)
    x = res_block(sd, 'diff_mid_block2', x, temb)
    if trajcontrol:
        x = x + ctrl[4]
    for lvl in (4, 3, 2, 1):                                       # trajnet.py:243-271
        x = up(sd, f'diff_upsample{lvl}', x)
        x = res_block(sd, f'diff_dec{lvl}', torch.cat([x, h_diff[lvl - 1]], dim=1), temb)
        if trajcontrol:
            x = x + ctrl[lvl - 1]
    x = conv_block(sd, 'diff_final_conv.0.block.', x)
    x = F.conv1d(x, sd['diff_final_conv.1.weight'], sd['diff_final_conv.1.bias'])
    return x.permute(0, 2, 1)
