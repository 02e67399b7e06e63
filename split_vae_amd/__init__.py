"""split_vae_amd -- MI355X-native SPLIT-VAE training path (drop-in for 51616/split-vae's vae/ hot path).

The compute lives in libsplitvae_hip.so (hand-written HIP for gfx950, C ABI in include/splitvae.h);
this package is the thin host-side mirror of the reference's Python surface
(augmentation.Augmentator, model.LGVae, trainer.train_step, main's flags).
"""
__version__ = "0.1.0"

import os as _os


def configure_hw_queues(n=3):
    """HIP maps streams onto at most GPU_MAX_HW_QUEUES hardware queues (default 4).  The single-GPU step uses the null stream and one
    side stream; the data-parallel step adds the communication stream -- and with a FOURTH hardware queue active every kernel of the
    step slows down on the MI355X boxes measured (512-image step through the RCCL path with one rank: 2.10 ms plain, 3.0 ms with 4
    queues, 2.7 with 8, 2.20 with 1 / 2 / 3; scripts/exp_dp_overhead.py).  Three queues give main / side /
    communication a queue each.

    The variable is process-wide and only read when the HIP runtime initialises, so this is NOT done at import: the entry points
    (main.main, spair_main.main, bench.py, dist.init_from_env) call it before they touch the GPU; an embedding application sets
    GPU_MAX_HW_QUEUES itself.  An existing setting is respected.  Returns the value in effect, warns when it is too
    late to have one."""
    if "GPU_MAX_HW_QUEUES" in _os.environ:
        return _os.environ["GPU_MAX_HW_QUEUES"]
    try:
        import torch
        late = torch.cuda.is_initialized()
    except Exception:
        late = False
    if late:
        import warnings
        warnings.warn("split_vae_amd: the HIP runtime is already initialised and GPU_MAX_HW_QUEUES is unset: the data-parallel step "
                      "runs ~35 % slower with the default 4 hardware queues (set GPU_MAX_HW_QUEUES=3 before the first torch.cuda call)")
        return None
    _os.environ["GPU_MAX_HW_QUEUES"] = str(n)
    return str(n)
