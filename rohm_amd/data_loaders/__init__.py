"""Device versions of the motion-representation helpers the drivers use between the diffusion stages."""
