"""split_vae_amd -- MI355X-native SPLIT-VAE training path (drop-in for 51616/split-vae's vae/ hot path).

The compute lives in libsplitvae_hip.so (hand-written HIP for gfx950, C ABI in include/splitvae.h);
this package is the thin host-side mirror of the reference's Python surface
(augmentation.Augmentator, model.LGVae, trainer.train_step, main's flags).
"""
__version__ = "0.1.0"

import os as _os

# HIP maps streams onto at most GPU_MAX_HW_QUEUES hardware queues (default 4).  The single-GPU step uses the null stream and one
# or two side streams; the data-parallel step adds the communication stream -- and with a FOURTH hardware queue active every
# kernel of the step slows down on the MI355X boxes measured (512-image step through the RCCL path with one rank: 2.10 ms plain,
# 3.0 ms with 4 queues, 2.7 with 8, 2.20 with 1 / 2 / 3; scripts/exp_dp_overhead.py).  Three queues give
# main / side / communication a queue each.  Only effective when set before the HIP runtime initialises (import this package --
# or set the variable -- before the first torch.cuda call); an existing setting is respected.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")
