"""Inference-side pieces of the VQ-VAE package that consume PSO results
(reference src/inverter/utils_vq_vae/)."""
