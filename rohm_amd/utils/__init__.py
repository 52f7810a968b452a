"""Host utilities: `model_util.create_gaussian_diffusion` (utils/model_util.py of the reference) and the
deterministic synthetic weights / body model / inputs used by tests, smoke and bench (`synth`)."""
