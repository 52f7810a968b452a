"""Phase timeline of the encoder-stack launch at the batch sizes of the BASELINE configs (B = 64: encoder_stack_kernel<4>, B = 32: <8>):
per-phase spans, meetings, and the rate of the attention phase inside the stack.  Text to stdout, JSON to the path given.

    python scripts/stack_timeline.py [out.json] [B ...]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from rohm_amd import stack_timeline                      # noqa: E402
from rohm_amd.model.posenet import PoseNet               # noqa: E402
from rohm_amd.utils import synth                         # noqa: E402


class DS:
    pose_feat_dim, traj_feat_dim = 272, 22


def main():
    out = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith('.json') else None
    sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [64, 32]
    dev = 'cuda:0'
    net = PoseNet(DS(), 294, latent_dim=512, ff_size=1024, num_layers=8, num_heads=4, traj_feat_dim=22,
                  body_model_path=torch.nn.Identity(), device=dev)
    net.load_state_dict(synth.posenet_state_dict(0), strict=True)
    net = net.to(dev).eval()
    recs = {}
    for B in sizes:
        rec = stack_timeline.measure(net, B, reps=8, device=dev)
        recs[f'b{B}'] = rec
        print(stack_timeline.text(rec))
        print()
    if out:
        with open(out, 'w') as f:
            json.dump(recs, f, indent=1)


if __name__ == '__main__':
    main()
