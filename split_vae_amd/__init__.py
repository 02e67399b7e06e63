"""split_vae_amd -- MI355X-native SPLIT-VAE training path (drop-in for 51616/split-vae's vae/ hot path).

The compute lives in libsplitvae_hip.so (hand-written HIP for gfx950, C ABI in include/splitvae.h);
this package is the thin host-side mirror of the reference's Python surface
(augmentation.Augmentator, model.LGVae, trainer.train_step, main's flags).
"""
__version__ = "0.1.0"
