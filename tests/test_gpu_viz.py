"""Inference surface + image grids on the device: encode / decode / encode_y and the canvases of
vae/visualizer.py for LGVae and LGGMVae."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
H = 32


def _batch(n, seed=0):
    from split_vae_amd import data
    from split_vae_amd.augmentation import Augmentator
    x = data.synthetic_images(n, H, H, seed=seed, device="cuda")
    return Augmentator("scramble", size=4, seed=1).augment(x)


@pytest.mark.parametrize("kind", ["lgvae", "lggmvae"])
def test_grids(lib_built, tmp_path, kind):
    from split_vae_amd import visualizer as viz
    from split_vae_amd.gm import LGGMVae
    from split_vae_amd.model import LGVae
    if kind == "lgvae":
        model = LGVae(128, 128, image_shape=[-1, H, H, 3], dtype="bf16", device="cuda", seed=2)
    else:
        model = LGGMVae(128, 128, [-1, H, H, 3], 30, 0.4, dtype="bf16", device="cuda", seed=2)
    out = str(tmp_path) + "/"
    c = viz.generate(model, filename="gen", filepath=out, seed=0)
    assert c.shape == (10 * H, 10 * H, 3) and np.isfinite(c).all() and c.min() >= 0 and c.max() <= 1
    assert viz.load_png(out + "gen.png").shape == (10 * H, 10 * H, 3)
    cx, ch = viz.generate_varying_latent(model, "lower", filename="lo", filepath=out, seed=1)
    cu = viz.generate_varying_latent(model, "upper", filename="up", filepath=out, seed=1)
    assert cx.shape == ch.shape == cu.shape == (10 * H, 10 * H, 3)
    # 'upper' fixes the local latent: decoder_x_hat sees one z_l, but the x grid varies with z_g
    assert np.abs(cu[:H, :H] - cu[:H, H:2 * H]).max() > 0
    # 'lower' fixes the global latent: every x_hat tile still differs (100 local draws)
    assert np.abs(ch[:H, :H] - ch[:H, H:2 * H]).max() > 0
    test_ds = [_batch(24)]
    rx, rh = viz.reconstruction_test_lg_vae(model, test_ds, label=False, filename="_t", filepath=out, n=10)
    assert rx.shape == rh.shape == (2 * H, 10 * H, 3)
    src = (test_ds[0][:10].float().cpu().numpy() + 1) * 0.5
    assert np.allclose(rx[H:, :H], src[0, :, :, :3]) and np.allclose(rh[H:, H:2 * H], src[1, :, :, 3:])   # row 1 = the inputs
    st = viz.style_transfer_celeba(model, test_ds, label=False, filename="_t", filepath=out, n=10)
    assert st.shape == (4 * H, 10 * H, 3)
    assert np.allclose(st[:H, :H], src[0, :, :, :3])
    for f in ("lo", "x_hat_lo", "up", "x_reconstruction_test_t", "x_hat_reconstruction_test_t", "style_transfer_celeba_t"):
        assert viz.load_png(out + f + ".png").ndim == 3


def test_decode_of_encode_is_the_forward_reconstruction(lib_built):
    """model.decode(*model.encode(x)) equals the x_mean of model(x) for the same eps (vae/model.py:204-218)."""
    from split_vae_amd.model import LGVae
    model = LGVae(128, 128, image_shape=[-1, H, H, 3], dtype="f32", device="cuda", seed=2)
    img = _batch(6)
    eps = (torch.randn(6, 128, device="cuda"), torch.randn(6, 128, device="cuda"))
    full = model(img, eps=eps)
    z_x, z_h = model.encode(img, eps=eps)
    torch.testing.assert_close(z_x, full[2], rtol=1e-5, atol=1e-5)      # two runs: the split-K heads sum in a different order
    torch.testing.assert_close(z_h, full[5], rtol=1e-5, atol=1e-5)
    x_mean, xh_mean = model.decode(z_x, z_h, rescale=False)
    torch.testing.assert_close(x_mean, full[0], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(xh_mean, full[6], rtol=1e-4, atol=1e-4)
    r, _ = model.decode(z_x, z_h, rescale=True)
    torch.testing.assert_close(r, torch.clamp((full[0] + 1) * 0.5, 0, 1), rtol=1e-4, atol=1e-4)


def test_cluster_grids(lib_built, tmp_path):
    from split_vae_amd import visualizer as viz
    from split_vae_amd.gm import LGGMVae
    model = LGGMVae(128, 128, [-1, H, H, 3], 30, 0.4, dtype="bf16", device="cuda", seed=2)
    out = str(tmp_path) + "/"
    for vary in ("zg", "zg_zl", "y_zg"):
        c = viz.generate_cluster(model, vary, filepath=out, seed=3)
        assert c.shape == (10 * H, 10 * H, 3) and np.isfinite(c).all()
    # 'zg_zl': rows share the global draw, columns share the local draw -> tiles differ along both axes
    c = viz.generate_cluster(model, "zg_zl", filename="g", filepath=out, seed=4)
    assert np.abs(c[:H, :H] - c[:H, H:2 * H]).max() > 0 and np.abs(c[:H, :H] - c[H:2 * H, :H]).max() > 0
    strips = viz.unseen_cluster_lg(model, [_batch(40), _batch(24, seed=5)], label=False, filename="_t", filepath=out)
    assert len(strips) >= 1 and all(v.shape[0] == H and v.shape[1] % H == 0 and v.shape[1] <= 7 * H for v in strips.values())
    y, logits = model.get_y(_batch(6)[..., :3].contiguous())
    assert y.shape == logits.shape == (6, 30) and torch.allclose(y.sum(1).cpu(), torch.ones(6), atol=1e-4)
