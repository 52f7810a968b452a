"""Drop-in for RoHM's `diffusion/gaussian_diffusion_trajnet.py` (module object passed as `gd=`)."""
from .ddpm import (DDPMSampler, LossType, ModelMeanType, ModelVarType, _extract_into_tensor,  # noqa: F401
                   betas_for_alpha_bar, get_named_beta_schedule)


class GaussianDiffusionTrajNet(DDPMSampler):
    """TrajNet diffusion: native 100-step cosine schedule, no guidance
    (gaussian_diffusion_trajnet.py:440-466)."""

    supports_guidance = False

    def eval_losses(self, model, batch, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                    device=None, progress=False, skip_timesteps=0, init_data=None, randomize_class=False,
                    cond_fn_with_grad=False, cond_grad_weight=1.0, dump_steps=None, const_noise=False,
                    cur_epoch=0, timestep_respacing='', compute_loss=True, smplx_model=None):
        """Entry point used by the drivers (gaussian_diffusion_trajnet.py:878-915) -> (loss report or None, x0 [B, T, 13])."""
        return self._eval(model, batch, shape, progress, clip_denoised, cond_fn_with_grad, None, False,
                          timestep_respacing, compute_loss, smplx_model)
