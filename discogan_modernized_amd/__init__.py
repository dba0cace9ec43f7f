"""discogan_modernized_amd -- MI355X (gfx950) native DiscoGAN training-step path.

Hand-written HIP kernels behind a C ABI (include/discogan_hip.h, csrc/), a ctypes binding, and a
Python host layer that mirrors the reference's module API (model.Generator / model.Discriminator,
get_gan_loss / get_fm_loss, the image_translation CLI).  No CPU fallback: ops raise without the
built library or without a HIP device.
"""
__version__ = "0.1.0"
