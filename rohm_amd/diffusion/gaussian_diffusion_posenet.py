"""Drop-in for RoHM's `diffusion/gaussian_diffusion_posenet.py` (module object passed as `gd=`).

Exposes `get_named_beta_schedule`, `ModelMeanType`, `ModelVarType`, `LossType` and
`GaussianDiffusionPoseNet`; the sampler itself is `rohm_amd.diffusion.ddpm.DDPMSampler`.
"""
from .ddpm import (DDPMSampler, LossType, ModelMeanType, ModelVarType, _extract_into_tensor,  # noqa: F401
                   betas_for_alpha_bar, get_named_beta_schedule)


class GaussianDiffusionPoseNet(DDPMSampler):
    """PoseNet diffusion: 1000-step cosine schedule, optional test-time guidance
    (`grad_type` in {None, 'amass', 'prox'}; gaussian_diffusion_posenet.py:436-480)."""

    supports_guidance = True

    def eval_losses(self, model, batch, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                    device=None, progress=False, skip_timesteps=0, init_data=None, randomize_class=False,
                    cond_fn_with_grad=False, grad_type=None, early_stop=False, cond_grad_weight=1.0,
                    dump_steps=None, const_noise=False, cur_epoch=0, timestep_respacing='', compute_loss=True,
                    smplx_model=None, epoch=0):
        """Entry point used by the drivers (gaussian_diffusion_posenet.py:913-962) ->
        (loss report or None, x0 [B, 294, 1, T])."""
        return self._eval(model, batch, shape, progress, clip_denoised, cond_fn_with_grad, grad_type, early_stop,
                          timestep_respacing, compute_loss, smplx_model, epoch)
