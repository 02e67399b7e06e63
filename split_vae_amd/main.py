"""CLI mirror of vae/main.py:15-31 (same flag names and defaults) for the SPLIT-VAE path.

    python -m split_vae_amd.main --beta 120 --patch_size 8 --dataset celeba64 -no_label --synthetic

Extra flags (not in the reference): --synthetic, --dtype, --seed, --log_every.
"""
import argparse

from .utils import dotdict


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('-viz', action='store_true')  # visualize results
    parser.add_argument('--global_latent_dims', type=int, nargs='?', default=128)
    parser.add_argument('--local_latent_dims', type=int, nargs='?', default=128)
    parser.add_argument('--learning_rate', type=float, nargs='?', default=1e-4)
    parser.add_argument('--beta', type=float, nargs='?', default=40)
    parser.add_argument('--dataset', type=str, nargs='?', default='svhn')
    parser.add_argument('--training_steps', type=int, nargs='?', default=1000000)
    parser.add_argument('--batch_size', type=int, nargs='?', default=64)
    parser.add_argument('--patch_size', type=int, nargs='?', default=1)
    parser.add_argument('--augmentation', type=str, nargs='?', default='scramble')
    parser.add_argument('-no_label', action='store_true')
    parser.add_argument('--model', type=str, nargs='?', default='lgvae')
    parser.add_argument('--y_size', type=int, nargs='?', default=30)
    parser.add_argument('--tau', type=float, nargs='?', default=0.4)
    parser.add_argument('--alpha', type=float, nargs='?', default=40)
    parser.add_argument('-allow_growth', action='store_true')
    # --- additions
    parser.add_argument('--synthetic', action='store_true', help='synthetic batches in the reference data domain')
    parser.add_argument('--dtype', type=str, default='bf16', choices=['bf16', 'f32'])
    parser.add_argument('--seed', type=int, default=0)
    parser.add_argument('--log_every', type=int, default=10000)
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    config = dotdict(vars(args))
    config.label = not config.no_label
    print('Config:', config)
    from . import data, trainer
    from .augmentation import Augmentator
    from .model import LGVae
    from .optimizer import Adam

    augmentor = Augmentator(type=config.augmentation, size=config.patch_size, seed=config.seed)
    train_ds, test_ds, input_shape = data.get_dataset(config.dataset, config.batch_size, synthetic=config.synthetic)
    if config.label and config.synthetic:
        config.label = False      # synthetic batches carry no labels
    train_ds = (augmentor.augment(x) for x in train_ds)                 # vae/main.py:57-61
    test_batches = [augmentor.augment(x) for x in test_ds]
    if args.model == 'lgvae':
        model = LGVae(global_latent_dims=config.global_latent_dims, local_latent_dims=config.local_latent_dims,
                      image_shape=input_shape, dtype=config.dtype, seed=config.seed)
        optimizer = Adam(learning_rate=config.learning_rate)
    elif args.model == 'lggmvae':                                       # vae/main.py:66-69
        from .gm import LGGMVae
        from .optimizer import ExponentialDecay
        lr_schedule = ExponentialDecay(config.learning_rate, decay_steps=1000000, decay_rate=0.4, staircase=True)
        optimizer = Adam(learning_rate=lr_schedule)
        model = LGGMVae(global_latent_dims=config.global_latent_dims, local_latent_dims=config.local_latent_dims,
                        image_shape=input_shape, y_size=config.y_size, tau=config.tau, dtype=config.dtype, seed=config.seed)
    else:
        raise NotImplementedError("--model %s: GMVae has no local branch and is outside the SPLIT path (SURVEY 8f)" % args.model)
    model.summary()
    print('Training local-global autoencoder')
    return trainer.train_local_global_autoencoder(model, optimizer, config.dataset, train_ds, test_batches, config=config)


if __name__ == "__main__":
    main()
