"""Drop-in modules for the reference's `diffusion` package: the DDPM / DDIM samplers over the fused HIP loops,
schedule tables in float64 like the reference, respacing."""
