"""CLI mirror of spair/main.py:19-50 (same flag names and defaults) for SPAIR / SPLIT-SPAIR.

    python -m split_vae_amd.spair_main --dataset cub_solid_fixed --z_bg_beta 10 --patch_size 8 --latent_size 64 --bg_latent_size 4 \\
        --local_latent_size 4 --model lg_spair -split_z_l -concat_z_what -dense_local -dense_bg --training_steps 200 --synthetic

The Multi-Bird canvases are synthesised by the reference from CUB mask blobs that are not in its repository (spair/data.py:14-15),
so the only data source here is --synthetic: 48x48x3 canvases in [0,1] (the shape get_cub_dataset reports, spair/data.py:258-278)
with 0-5 soft-edged blobs on a solid background and the blob count as the label.
Extra flags (not in the reference): --synthetic, --seed, --log_every, --graph, --dtype.
"""
import argparse

import torch

from .utils import dotdict

REFERENCE_SWITCHES = ["-no_label", "-allow_growth", "-split_z_l", "-dense_bg", "-dense_local", "-concat_bg", "-concat_z_what",
                      "-concat_backbone"]
REFERENCE_OPTIONS = [
    ("--learning_rate", float, 1e-4), ("--beta", float, 0.5), ("--dataset", str, "cub_solid_fixed"), ("--channel", int, 3),
    ("--training_steps", int, 100000), ("--batch_size", int, 32), ("--runs", int, 1), ("--tau", float, 0.8), ("--object_size", int, 32),
    ("--latent_size", int, 128), ("--anneal_until", float, 1.0), ("--z_pres_anneal_step", float, 10000.0), ("--prior_z_zoom", float, 0.0),
    ("--prior_z_zoom_start", float, 10.0), ("--reconstruction_weight", float, 1.0), ("--bg_latent_size", int, 4),
    ("--local_latent_size", int, 64), ("--z_bg_beta", float, 10.0), ("--z_l_beta", float, 0.1), ("--z_what_beta", float, 0.1),
    ("--model", str, "spair"), ("--patch_size", int, 4), ("--augmentation", str, "scramble"),
]


def build_parser():
    ap = argparse.ArgumentParser(description="SPAIR / SPLIT-SPAIR training on MI355X (flags of the reference's spair/main.py)")
    for sw in REFERENCE_SWITCHES:
        ap.add_argument(sw, action="store_true")
    for flag, ty, default in REFERENCE_OPTIONS:
        ap.add_argument(flag, type=ty, nargs="?", default=default)
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--log_every", type=int, default=1000)
    ap.add_argument("--dtype", type=str, default="f32", choices=["f32", "bf16"],
                    help="bf16: the spatial convolutions on the bf16 MFMA kernels (fp32 accumulation / master weights); f32 = the reference's precision")
    ap.add_argument("--save_weights", action="store_true", help="write models/<time>.h5 (Keras HDF5 weights) at the end, as spair/trainer.py:424 does")
    ap.add_argument("--clipnorm_semantics", type=str, default="tf2.0", choices=["tf2.0", "tf2.4"],
                    help="Adam(clipnorm=1.0) in a tape.gradient -> apply_gradients loop (spair/main.py:109, spair/trainer.py:226-227): the pinned "
                         "tensorflow_gpu==2.0.0 does not clip in apply_gradients (default); TF >= 2.4 clips every gradient tensor with tf.clip_by_norm")
    ap.add_argument("--graph", action="store_true", help="capture the train step into a hipGraph and replay it (spair_trainer.GraphedTrainStep)")
    return ap


def default_config(**kw):
    c = dotdict(vars(build_parser().parse_args([])))
    c.image_size, c.test_size = [48, 48, 3], [48, 48, 3]
    c.update(kw)
    c.label = not c.no_label
    return c


def synthetic_canvases(B, seed=0, device="cuda", size=48):
    """[B,size,size,3] in [0,1] + the object count per canvas."""
    g = torch.Generator().manual_seed(seed)
    ys = torch.arange(size).view(1, size, 1).float()
    xs = torch.arange(size).view(1, 1, size).float()
    bg = torch.rand(B, 1, 1, 3, generator=g) * 0.5
    img = bg.expand(B, size, size, 3).clone()
    count = torch.randint(0, 6, (B,), generator=g)
    for k in range(5):
        on = (count > k).float().view(B, 1, 1, 1)
        cy, cx = torch.rand(B, 1, 1, generator=g) * size, torch.rand(B, 1, 1, generator=g) * size
        r = 3.0 + torch.rand(B, 1, 1, generator=g) * 5.0
        col = 0.5 + 0.5 * torch.rand(B, 1, 1, 3, generator=g)
        a = torch.sigmoid((r - torch.sqrt((ys - cy) ** 2 + (xs - cx) ** 2)) * 1.5).unsqueeze(-1) * on
        img = img * (1 - a) + col * a
    return img.clamp(0, 1).to(device), count.float().to(device)


def main(argv=None):
    args = build_parser().parse_args(argv)
    from . import configure_hw_queues
    configure_hw_queues()                            # before the first HIP call (split_vae_amd/__init__.py)
    config = dotdict(vars(args))
    config.label = not config.no_label
    print('Config:', config)
    if not config.synthetic:
        raise SystemExit("the Multi-Bird source blobs are not in the reference repository (spair/data.py:14-15): pass --synthetic")
    from . import spair, spair_trainer
    from .augmentation import Augmentator
    config.image_size, config.test_size = [48, 48, config.channel], [48, 48, config.channel]
    augmentor = Augmentator(type=config.augmentation, size=config.patch_size, seed=config.seed)
    lg = config.model == 'lg_spair'

    def batches():
        i = 0
        while True:
            x, _ = synthetic_canvases(config.batch_size, seed=config.seed + 1 + i)
            yield augmentor.augment(x) if lg else x                     # spair/main.py:71-72
            i += 1

    tx, ty = synthetic_canvases(config.batch_size, seed=config.seed + 10 ** 6)
    test_batches = [[(augmentor.augment(tx) if lg else tx, ty) if config.label else (augmentor.augment(tx) if lg else tx)]]
    history = None
    for _ in range(args.runs):
        print('Creating model...')
        model = spair.get_model(config, seed=config.seed)
        print(type(model))
        model.summary()
        optimizer = spair_trainer.ClipnormAdam(config.learning_rate, clipnorm=1.0,         # spair/main.py:109
                                               clip_in_apply=(config.clipnorm_semantics == "tf2.4"))
        print('Training SPAIR')
        history = spair_trainer.train_spair(model, optimizer, config.dataset, batches(), test_batches, config)
    return history


if __name__ == "__main__":
    main()
