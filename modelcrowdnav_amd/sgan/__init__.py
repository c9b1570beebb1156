"""Social-GAN trajectory generator: weight container + packed HIP inference (sgan_step.hip)."""
