"""The zero-edit drop-in (rohm_amd/dropin.py): after `install()` the LITERAL import block of the RoHM drivers
(test_amass_full.py:10-15 = test_prox_egobody.py:10-15) resolves to rohm_amd, the drivers' own construction calls
(test_amass_full.py:132-188) build the three networks and the three diffusions, and -- on the GPU -- the drivers' own
`eval_losses(...)` call (test_amass_full.py:376-384) lands on the reference's 8-step result.

Every case runs in a child interpreter: the aliases live in `sys.modules`, and this test session also imports the real reference
under the same names (oracle/refload.py)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'

# the six statements of the drivers' import block that name the hot path (API surface: module paths and class names)
IMPORT_BLOCK = """\
from model.posenet import PoseNet
from diffusion import gaussian_diffusion_posenet
from model.trajnet import TrajNet
from diffusion import gaussian_diffusion_trajnet
from diffusion.respace import SpacedDiffusionPoseNet, SpacedDiffusionTrajNet
from utils.model_util import create_gaussian_diffusion
"""

# the drivers' construction statements with their exact keyword arguments (test_amass_full.py:132-188)
CONSTRUCT = """\
model_posenet = PoseNet(dataset=test_pose_dataset, body_feat_dim=test_pose_dataset.body_feat_dim,
                        latent_dim=512, ff_size=1024, num_layers=8, num_heads=4, dropout=0.1, activation="gelu",
                        body_model_path=args.body_model_path,
                        device=dist_util.dev(),
                        traj_feat_dim=test_pose_dataset.traj_feat_dim,
                        ).to(dist_util.dev())
model_posenet.load_state_dict(weights_posenet)
model_posenet.eval()
diffusion_posenet_eval = create_gaussian_diffusion(args, gd=gaussian_diffusion_posenet,
                                                   return_class=SpacedDiffusionPoseNet,
                                                   num_diffusion_timesteps=args.diffusion_steps_posenet,
                                                   timestep_respacing=args.timestep_respacing_eval,
                                                   device=dist_util.dev())
model_trajnet = TrajNet(time_dim=32, mid_dim=512,
                cond_dim=test_traj_dataset.traj_feat_dim,
                traj_feat_dim=test_traj_dataset.traj_feat_dim,
                trajcontrol=False,
                device=dist_util.dev(),
                dataset=test_traj_dataset,
                repr_abs_only=args.repr_abs_only,
                ).to(dist_util.dev())
model_trajnet_control = TrajNet(time_dim=32, mid_dim=512,
                        cond_dim=test_traj_dataset.traj_feat_dim,
                        traj_feat_dim=test_traj_dataset.traj_feat_dim,
                        trajcontrol=True,
                        device=dist_util.dev(),
                        dataset=test_traj_dataset,
                        repr_abs_only=args.repr_abs_only,
                        ).to(dist_util.dev())
model_trajnet.load_state_dict(weights_trajnet)
model_trajnet.eval()
model_trajnet_control.load_state_dict(weights_trajnet_control)
model_trajnet_control.eval()
diffusion_trajnet_eval = create_gaussian_diffusion(args, gd=gaussian_diffusion_trajnet,
                                                   return_class=SpacedDiffusionTrajNet,
                                                   num_diffusion_timesteps=args.diffusion_steps_trajnet,
                                                   timestep_respacing=args.timestep_respacing_eval,
                                                   device=dist_util.dev())
diffusion_trajnet_control_eval = create_gaussian_diffusion(args, gd=gaussian_diffusion_trajnet,
                                                           return_class=SpacedDiffusionTrajNet,
                                                           num_diffusion_timesteps=args.diffusion_steps_trajnet,
                                                           timestep_respacing=args.timestep_respacing_eval,
                                                           device=dist_util.dev())
"""

# the driver's PoseNet sampling call (test_amass_full.py:376-384)
EVAL_CALL = """\
_, val_output_pose = diffusion_posenet_eval.eval_losses(model=model_posenet, batch=test_batch_pose,
                                                        shape=shape, progress=True,
                                                        clip_denoised=False,
                                                        timestep_respacing=args.timestep_respacing_eval,
                                                        cond_fn_with_grad=args.cond_fn_with_grad,
                                                        early_stop=args.early_stop,
                                                        compute_loss=False,
                                                        grad_type='amass',
                                                        smplx_model=smplx_neutral)
"""

PRELUDE = """\
import sys, types, warnings
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + '/tests')
{extra_path}
warnings.simplefilter('ignore')
import torch
import rohm_amd.dropin as dropin
dropin.install()
"""

SETUP = """\
from rohm_amd.utils import synth
dev = torch.device({dev!r})
dist_util = types.SimpleNamespace(dev=lambda: dev)           # utils/dist_util.py:45-52 (reference host code, stays the reference's)
args = types.SimpleNamespace(body_model_path='body_models/smplx_model', diffusion_steps_posenet={steps}, diffusion_steps_trajnet=100,
                             timestep_respacing_eval='', noise_schedule='cosine', sigma_small=True, repr_abs_only=True,
                             cond_fn_with_grad=False, early_stop=False)
mean, std = synth.synthetic_stats(0)
test_pose_dataset = types.SimpleNamespace(body_feat_dim=294, pose_feat_dim=272, traj_feat_dim=22, joints_num=22, Mean=mean, Std=std)
test_traj_dataset = types.SimpleNamespace(traj_feat_dim=13, pose_feat_dim=272, Mean=mean, Std=std)
weights_posenet = synth.posenet_state_dict({wseed})
weights_trajnet = synth.trajnet_state_dict(71, trajcontrol=False)
weights_trajnet_control = synth.trajnet_state_dict(72, trajcontrol=True)
"""


def _run(code, timeout=600):
    env = dict(os.environ, PYTHONPATH='')
    p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=timeout, cwd='/tmp', env=env)
    assert p.returncode == 0, p.stdout[-3000:] + '\n' + p.stderr[-3000:]
    return p.stdout


def _prelude(with_reference):
    extra = f'sys.path.insert(0, {REF!r})' if with_reference else ''
    return PRELUDE.format(root=ROOT, extra_path=extra)


CHECKS = """\
import model.posenet, model.trajnet, diffusion.respace, utils.model_util
assert PoseNet.__module__ == 'rohm_amd.model.posenet' and TrajNet.__module__ == 'rohm_amd.model.trajnet'
assert gaussian_diffusion_posenet.__name__ == 'rohm_amd.diffusion.gaussian_diffusion_posenet'
assert gaussian_diffusion_trajnet.__name__ == 'rohm_amd.diffusion.gaussian_diffusion_trajnet'
assert SpacedDiffusionPoseNet.__module__ == SpacedDiffusionTrajNet.__module__ == 'rohm_amd.diffusion.respace'
assert create_gaussian_diffusion.__module__ == 'rohm_amd.utils.model_util'
assert model.posenet.PoseNet is PoseNet and dropin.installed()
"""


@pytest.mark.parametrize('with_reference', [False, True])
def test_literal_import_block_and_driver_construction(with_reference):
    """Without a RoHM checkout on sys.path (parents synthesised) and with one (the checkout's `utils` / `diffusion` / `model`
    packages stay the parents: `utils.dist_util`, `utils.fixseed`, `diffusion.logger` keep coming from the checkout)."""
    if with_reference and not os.path.isdir(os.path.join(REF, 'model')):
        pytest.skip('no reference checkout in this environment')
    code = _prelude(with_reference) + IMPORT_BLOCK + CHECKS + SETUP.format(dev='cpu', steps=1000, wseed=3) + CONSTRUCT + textwrap.dedent("""\
        assert len(model_posenet.state_dict()) >= 108 and diffusion_posenet_eval.num_timesteps == 1000
        assert diffusion_trajnet_eval.num_timesteps == diffusion_trajnet_control_eval.num_timesteps == 100
        assert model_trajnet_control.trajcontrol and not model_trajnet.trajcontrol
        """)
    if with_reference:
        code += textwrap.dedent(f"""\
            from utils import dist_util as real_dist_util
            from utils.fixseed import fixseed
            import diffusion.logger
            assert real_dist_util.__file__.startswith({REF!r}) and fixseed.__module__ == 'utils.fixseed'
            assert diffusion.logger.__file__.startswith({REF!r})
            # and the literal lines of the driver file itself
            lines = open({REF!r} + '/test_amass_full.py').read().split('\\n')[9:15]
            assert [l.strip() for l in lines] == [l.strip() for l in {IMPORT_BLOCK!r}.strip().split('\\n')], lines
            exec('\\n'.join(lines))
            lines = open({REF!r} + '/test_prox_egobody.py').read().split('\\n')[9:15]
            exec('\\n'.join(lines))
            """)
    code += textwrap.dedent("""\
        dropin.uninstall()
        assert 'model.posenet' not in sys.modules or sys.modules['model.posenet'].__name__ != 'rohm_amd.model.posenet'
        print('OK')
        """)
    assert 'OK' in _run(code)


def test_runner_executes_a_driver_script_unmodified(tmp_path):
    """`python -m rohm_amd.dropin driver.py args...`: the script sees the aliases, its own directory first on sys.path, its argv."""
    script = tmp_path / 'driver.py'
    script.write_text(IMPORT_BLOCK + textwrap.dedent("""\
        import sys, os
        import sibling
        assert sys.argv[1:] == ['--config', 'x.yaml'], sys.argv
        assert PoseNet.__module__ == 'rohm_amd.model.posenet' and __name__ == '__main__'
        print('DRIVER-OK', sibling.VALUE)
        """))
    (tmp_path / 'sibling.py').write_text('VALUE = 7\n')
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, '-m', 'rohm_amd.dropin', str(script), '--config', 'x.yaml'], capture_output=True, text=True,
                       timeout=300, cwd='/tmp', env=env)
    assert p.returncode == 0 and 'DRIVER-OK 7' in p.stdout, p.stdout + p.stderr


@pytest.mark.gpu
def test_driver_eval_losses_call_through_the_dropin_vs_reference_golden():
    """The drivers' exact sampling call on the HIP networks, reached through the aliased import block, against the reference's own
    8-step run (tests/golden/posenet_loop8.npz)."""
    code = _prelude(False) + IMPORT_BLOCK + textwrap.dedent("""\
        from helpers import cpu_noise_sequence, golden, seeded, max_abs
        g = golden('posenet_loop8.npz')
        """) + SETUP.format(dev='cuda:0', steps='int(g["steps"])', wseed='int(g["weight_seed"])') + textwrap.dedent("""\
        test_pose_dataset.Mean, test_pose_dataset.Std = mean * 0, std * 0 + 1
        args.body_model_path = torch.nn.Identity()
        """) + CONSTRUCT + textwrap.dedent("""\
        smplx_neutral = None
        cond = seeded(int(g['cond_seed']), 2, 294, 1, 143)
        x_T, noises = cpu_noise_sequence(int(g['torch_seed']), (2, 294, 1, 143), int(g['steps']))
        diffusion_posenet_eval.noise_source = lambda step, like: (x_T if step == -1 else noises[step])
        test_batch_pose = {'cond': cond.to(dev), 'motion_repr_clean': torch.zeros(2, 294, 1, 143, device=dev)}
        shape = list(test_batch_pose['motion_repr_clean'].shape)
        """) + EVAL_CALL + textwrap.dedent("""\
        err = max_abs(val_output_pose.cpu(), torch.from_numpy(g['y']))
        print('max|HIP - reference| =', err)
        assert err < 1e-3, err
        # TrajNet through the same block: one forward of each network runs
        xt = torch.zeros(2, 144, 13, device=dev)
        t = torch.tensor([5, 50], device=dev)
        assert tuple(model_trajnet({'x_t': xt, 'cond': xt}, t).shape) == (2, 144, 13)
        print('OK')
        """)
    assert 'OK' in _run(code)
