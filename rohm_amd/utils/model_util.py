"""`create_gaussian_diffusion` factory (drop-in for RoHM's `utils/model_util.py:6-40`)."""
from ..diffusion.respace import space_timesteps


def create_gaussian_diffusion(args, gd, return_class, num_diffusion_timesteps=100, timestep_respacing='',
                              device='', dataset=None):
    """x0-prediction, fixed variance (small if `args.sigma_small`), MSE loss tag, no learned sigma."""
    steps = num_diffusion_timesteps
    betas = gd.get_named_beta_schedule(args.noise_schedule, steps, 1.)
    var_type = gd.ModelVarType.FIXED_SMALL if args.sigma_small else gd.ModelVarType.FIXED_LARGE
    return return_class(
        use_timesteps=space_timesteps(steps, timestep_respacing or [steps]),
        betas=betas,
        model_mean_type=gd.ModelMeanType.START_X,
        model_var_type=var_type,
        loss_type=gd.LossType.MSE,
        rescale_timesteps=False,
        dataset=dataset,
        device=device,
    )
