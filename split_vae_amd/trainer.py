"""Host-side mirror of vae/trainer.py for the SPLIT-VAE path: train_step_lg_vae / test_step_lg_vae
(vae/trainer.py:120-144, :199-233) and the loop of train_local_global_autoencoder (:72-421).

The step itself (forward, ELBO, backward, Keras-Adam, the 5 running means of :140-144) is one
native call into libsplitvae_hip.so; with torch.distributed initialised the call is split at the
phase boundaries so the RCCL gradient all-reduce overlaps the remaining backward work.
"""
import os
import time
from datetime import datetime

import torch

from . import dist as svdist
from ._lib import (PHASE_ADAM, PHASE_ALL, PHASE_BUCKET_EVENTS, PHASE_BWD_DECODERS, PHASE_BWD_ENC_CONVS, PHASE_BWD_ENC_HEADS,
                   PHASE_FORWARD, PHASE_INPUTS_STAGED, PHASE_LOSS, PHASE_NO_RECON, PHASE_PREP)
from .model import LGVae

METRIC_NAMES = ["x_recon_loss", "x_kl_loss", "x_hat_recon_loss", "x_hat_kl_loss", "total_kl_loss"]


class StepMetrics:
    """The five tf.keras.metrics.Mean of vae/trainer.py:99-113 that train_step_lg_vae updates;
    sums and count live on the device and are read back only by result()."""

    def __init__(self, plan):
        self.plan = plan

    def _acc(self):
        return self.plan.buffer("metric_acc", torch.float32, (8,))

    def result(self):
        a = self._acc().cpu()
        n = max(float(a[5]), 1.0)
        return {k: float(a[i]) / n for i, k in enumerate(METRIC_NAMES)}

    def reset_states(self, names=None):
        a = self._acc()
        if names is None:
            a.zero_()
        else:   # the reference resets only some of its means (vae/trainer.py:405-414)
            for n in names:
                a[METRIC_NAMES.index(n)] = 0


def last_losses(plan):
    """x_recon, x_kl, x_hat_recon, x_hat_kl, total_kl, total of the most recent step (device -> host)."""
    l = plan.buffer("losses", torch.float32, (8,)).cpu()
    return {k: float(l[i]) for i, k in enumerate(METRIC_NAMES + ["total_loss"])}


def _check_images(model, images):
    """The kernels read B*H*W*6 floats from the pointer: refuse anything that is not a [B,H,W,6] fp32 device batch
    (e.g. one image of a batch taken by a stray `train_data[0]`)."""
    if not torch.is_tensor(images) or images.dim() != 4 or tuple(images.shape[1:]) != (model.H, model.W, 6):
        raise ValueError("images must be a [B, %d, %d, 6] tensor (x | x_hat on the channel axis), got %s" %
                         (model.H, model.W, tuple(images.shape) if torch.is_tensor(images) else type(images)))
    if images.dtype != torch.float32 or not images.is_cuda or not images.is_contiguous():
        raise ValueError("images must be contiguous fp32 on the HIP device")


_DP_TWO_BUCKETS_MAX = int(os.environ.get("SV_DP_TWO_BUCKETS_MAX", "256"))     # per-GPU batch up to which the data-parallel step uses two buckets


def train_step(model, images, optimizer, eps=None, reducer=None, sample_offset=0, accumulate_metrics=True, keep_recon=True):
    """train_step_lg_vae (vae/trainer.py:120-144): forward, total = recon_x + recon_x_hat +
    beta*KL, gradients of the 40 variables, Adam update, metric update.  `images` [B,H,W,6] fp32
    on the device.  eps=(eps_x, eps_x_hat) pins the Sampling noise; default = Philox stream.
    keep_recon=False: the reference's step returns nothing and the loss is evaluated inside the head conv, so the
    reconstruction tensors (plan buffers out6_x / out6_xh) are dead and are not stored (SV_PHASE_NO_RECON); the training
    loop and bench.py run this way.  Losses, gradients and the update are identical either way."""
    from .gm import LGGMVae
    if isinstance(model, LGGMVae):
        raise TypeError("LGGMVae trains with gm.train_step_lg_gm_vae (vae/trainer.py:297-299 picks the step by model class)")
    if not isinstance(model, LGVae):
        raise NotImplementedError("GMVae (no local branch) is outside the SPLIT path (SURVEY 8f)")
    _check_images(model, images)
    B = images.shape[0]
    plan = model.plan(B)
    m, v = optimizer.slots(model.flat)
    lr = optimizer.lr()
    optimizer.iterations += 1
    ex, eh = (None, None) if eps is None else eps
    kw = dict(params=model.flat, grads=model.grad_flat, adam_m=m, adam_v=v, images6=images, eps_x=ex, eps_x_hat=eh,
              seed=model.seed, step=model._calls, sample_offset=sample_offset, lr=lr, beta1=optimizer.beta_1,
              beta2=optimizer.beta_2, adam_eps=optimizer.epsilon, t=optimizer.iterations,
              accumulate_metrics=accumulate_metrics)
    model._calls += 1
    nr = 0 if keep_recon else PHASE_NO_RECON
    if getattr(images, "_sv_staged_plan", None) is plan:      # Augmentator.scramble(..., plan=plan) filled in8_x / in8_xh already
        # ... and nothing has overwritten them since (generation), nor was `images` edited in place (version counter)
        if images._sv_staged_gen == plan.in8_gen and images._sv_staged_version == images._version:
            nr |= PHASE_INPUTS_STAGED
        images._sv_staged_plan = None                         # one step per staging
    if reducer is None or (reducer.world == 1 and not getattr(reducer, "force", False)):
        plan.step(PHASE_ALL | nr, **kw)
        return plan
    mode = getattr(reducer, "mode", "events")
    if mode == "auto":
        # one rank (SV_DIST_FORCE: the data-parallel path on a single device) has no link time to hide and one all-reduce is the cheapest hand-over
        # (profiles/r05_dp_ab.txt); with real peers the decoders' 16.4 MB travel beside the encoders' backward
        mode = "single" if reducer.world == 1 else "events"
    if mode == "events" and getattr(plan, "graph_on", False):
        mode = "single"          # captured steps record no bucket events (sv_lgvae_bucket_wait would return SV_E_STATE): one all-reduce behind the compute stream
    if mode == "single":
        # one all-reduce of the whole gradient buffer between the backward and Adam: nothing overlaps, but there is one
        # cross-stream hand-over instead of four and the backward runs as in the single-GPU step
        plan.step((PHASE_ALL & ~PHASE_ADAM) | nr, **kw)
        reducer.launch_all(model.grad_flat)
        reducer.wait()
        plan.step(PHASE_ADAM, grad_scale=reducer.grad_scale, **kw)
        return plan
    if mode == "events":
        # data parallel, default: ONE call for forward + loss + the whole backward (the single-GPU stream placement: weight gradients on the side
        # stream(s) beside the input-gradient chain, no join between the decoders' and the encoders' backward); the library records an event set as each
        # gradient bucket completes and the all-reduces are ordered behind THOSE (sv_lgvae_bucket_wait), so the decoders' bucket travels while the
        # encoders' backward runs.  Two host calls into the library per step instead of four or five.
        plan.step((PHASE_ALL & ~PHASE_ADAM) | PHASE_BUCKET_EVENTS | nr, **kw)
        reducer.launch(model.grad_flat, "decoders", after=(plan, 0))
        if B <= _DP_TWO_BUCKETS_MAX:
            reducer.launch(model.grad_flat, "encoders", after=(plan, 3))
        else:
            reducer.launch(model.grad_flat, "enc_heads", after=(plan, 1))
            reducer.launch(model.grad_flat, "enc_convs", after=(plan, 2))
        reducer.wait()
        plan.step(PHASE_ADAM, grad_scale=reducer.grad_scale, **kw)
        return plan
    # SV_DP_MODE=overlap (rounds 1-4): the step split at the phase boundaries, each bucket's all-reduce launched as soon as the phase that fills it is enqueued
    plan.step(PHASE_PREP | PHASE_FORWARD | PHASE_LOSS | PHASE_BWD_DECODERS | nr, **kw)
    reducer.launch(model.grad_flat, "decoders")
    if B <= _DP_TWO_BUCKETS_MAX:
        # shards of the global batch 512 (256 / 128 / 64 images per GPU): the encoders' backward is 0.15-0.2 ms of GPU time, less than the host
        # needs to enqueue two more phases and five more all-reduce calls (~30 us each); one encoder phase + one contiguous bucket.  One rank,
        # no link time: 0.740 -> 0.707 ms at 64 images, 0.916 -> 0.866 at 128, 1.250 -> 1.189 at 256 (profiles/r04_dp_one_rank.txt)
        plan.step(PHASE_BWD_ENC_HEADS | PHASE_BWD_ENC_CONVS, **kw)
        reducer.launch(model.grad_flat, "encoders")
    else:
        plan.step(PHASE_BWD_ENC_HEADS, **kw)
        reducer.launch(model.grad_flat, "enc_heads")
        plan.step(PHASE_BWD_ENC_CONVS, **kw)
        reducer.launch(model.grad_flat, "enc_convs")
    reducer.wait()
    plan.step(PHASE_ADAM, grad_scale=reducer.grad_scale, **kw)
    return plan


def test_step(model, images, labels=None, eps=None):
    """test_step_lg_vae (vae/trainer.py:199-233) without the classifier branches (:213-226: the
    probe classifier's weights are not in the reference repo, .MISSING_LARGE_BLOBS:1)."""
    if labels is not None:
        raise NotImplementedError("classifier-based metrics need svhn_classifier_weights.h5 (missing upstream)")
    _check_images(model, images)
    B = images.shape[0]
    plan = model.plan(B)
    ex, eh = (None, None) if eps is None else eps
    plan.step(PHASE_PREP | PHASE_FORWARD | PHASE_LOSS, params=model.flat, images6=images, eps_x=ex, eps_x_hat=eh,
              seed=model.seed ^ 0x7e57, step=model._calls)
    model._calls += 1
    return last_losses(plan)


test_step.__test__ = False   # not a pytest test


def train_local_global_autoencoder(model, optimizer, dataset, train_dataset, test_dataset, config):
    """Loop of vae/trainer.py:72-421 for LGVae: train; every 10 000 steps (incl. step 0) evaluate
    on the test set and print the reference's report; stop after training_steps; save weights.
    The image grids of vae/visualizer.py are written by _write_grids (visualizer.py)."""
    from . import gm
    if isinstance(model, gm.LGGMVae):               # vae/trainer.py:294-302: the step functions follow the model class
        return _train_lggmvae(model, optimizer, train_dataset, test_dataset, config)
    RUN_NAME = datetime.now().strftime("%Y%m%d-%H%M%S")
    model.beta = float(config.beta)
    os.makedirs("models", exist_ok=True)
    metrics = None
    start = time.time()
    log_every = int(config.get("log_every") or 10000)
    for step, train_data in enumerate(train_dataset):
        images = train_data[0] if config.label else train_data
        plan = train_step(model, images, optimizer, keep_recon=False)
        if metrics is None:
            metrics = StepMetrics(plan)
        if step % log_every == 0:
            torch.cuda.synchronize()
            print('Training time: {:.2f}'.format(time.time() - start))
            start = time.time()
            sums, n = dict.fromkeys(METRIC_NAMES, 0.0), 0
            for test_data in test_dataset:
                timg = test_data[0] if config.label else test_data
                l = test_step(model, timg)
                for k in METRIC_NAMES:
                    sums[k] += l[k]
                n += 1
            te = {k: sums[k] / max(n, 1) for k in METRIC_NAMES}
            print('Testing time: {:.2f}'.format(time.time() - start))
            tr = metrics.result()
            template = ('Training step {}\n'
                        '            X Recon Loss: {:.4f}, X KLD loss: {:.4f}, Total X loss: {:.4f} \n'
                        '            X hat Recon Loss: {:.4f}, X hat KLD loss: {:.4f}, Total X hat loss: {:.4f} \n'
                        '            Test X Recon Loss: {:.4f}, Test X KLD loss: {:.4f}, Test Total X loss: {:.4f} \n'
                        '            Test X hat Recon Loss: {:.4f}, Test X hat KLD loss: {:.4f}, Test Total X hat loss: {:.4f}\n'
                        '            Total KL train loss: {:.4f}, Total KL test loss: {:.4f}')
            print(template.format(step, tr["x_recon_loss"], tr["x_kl_loss"], tr["x_recon_loss"] + tr["x_kl_loss"],
                                  tr["x_hat_recon_loss"], tr["x_hat_kl_loss"], tr["x_hat_recon_loss"] + tr["x_hat_kl_loss"],
                                  te["x_recon_loss"], te["x_kl_loss"], te["x_recon_loss"] + te["x_kl_loss"],
                                  te["x_hat_recon_loss"], te["x_hat_kl_loss"], te["x_hat_recon_loss"] + te["x_hat_kl_loss"],
                                  tr["total_kl_loss"], te["total_kl_loss"]))
            _write_grids(model, test_dataset, config, os.path.join("output", RUN_NAME), step)
            # vae/trainer.py:405-414 resets x_recon / x_kl / total_kl but never the x_hat_* means
            metrics.reset_states(["x_recon_loss", "x_kl_loss", "total_kl_loss"])
            start = time.time()
        if step >= config.training_steps:
            print('Training done!')
            break
    return _save(model, RUN_NAME)


def _save(model, run_name):
    """vae/trainer.py:421: model.save_weights('models/'+RUN_NAME+'.h5') -- Keras HDF5 when libhdf5 is present, else .npz."""
    from . import h5io
    return model.save_weights('models/' + run_name + ('.h5' if h5io.available() else '.npz'))


def _write_grids(model, test_dataset, config, run_dir, step):
    """The image grids the reference writes at every evaluation (vae/trainer.py:385-396; vae/visualizer.py)."""
    from . import visualizer
    out = run_dir + "/"
    tag = "_it_" + str(step)
    visualizer.generate(model, filename="generate_it_" + str(step), filepath=out)
    first = next(iter(test_dataset), None)
    if first is not None:
        visualizer.reconstruction_test_lg_vae(model, test_dataset, label=config.label, filename=tag, filepath=out)
    visualizer.generate_varying_latent(model, vary="lower", filename="vary_lower_it_" + str(step), filepath=out)
    visualizer.generate_varying_latent(model, vary="upper", filename="vary_upper_it_" + str(step), filepath=out)
    svhn_file = os.path.join("data", "SVHN", "test_32x32.mat")
    if config.dataset == "svhn" and os.path.exists(svhn_file):
        visualizer.style_transfer_test(model, test_dataset, label=config.label, filename=tag, filepath=out)
    elif first is not None and (first[0] if config.label else first).shape[0] >= 20:
        visualizer.style_transfer_celeba(model, test_dataset, label=config.label, filename=tag, filepath=out)
    from .gm import LGGMVae
    if config.viz and isinstance(model, LGGMVae):                      # vae/trainer.py:398-403
        visualizer.unseen_cluster_lg(model, test_dataset, label=config.label, filename=tag, filepath=out)
        visualizer.generate_cluster(model, vary="zg", filename="generate_cluster_fix_zl_it_" + str(step), filepath=out)
        visualizer.generate_cluster(model, vary="zg_zl", filename="generate_cluster_it_" + str(step), filepath=out)
        visualizer.generate_cluster(model, vary="y_zg", filename="generate_multi_cluster_it_" + str(step), filepath=out)


def _train_lggmvae(model, optimizer, train_dataset, test_dataset, config):
    """The same loop for LGGMVae (train_step_lg_gm_vae / test_step_lg_gm_vae, vae/trainer.py:146-173, :235-272): the five
    training means + y_kl, evaluated every `log_every` steps; cluster accuracy needs labels and the probe classifier
    (missing upstream) and is not reported."""
    from . import gm
    RUN_NAME = datetime.now().strftime("%Y%m%d-%H%M%S")
    model.beta, model.alpha = float(config.beta), float(config.alpha)
    os.makedirs("models", exist_ok=True)
    acc, n_acc = None, 0
    start = time.time()
    log_every = int(config.get("log_every") or 10000)
    for step, train_data in enumerate(train_dataset):
        images = train_data[0] if config.label else train_data
        m = gm.train_step_lg_gm_vae(model, images, optimizer)
        acc = m.clone() if acc is None else acc + m
        n_acc += 1
        if step % log_every == 0:
            torch.cuda.synchronize()
            print('Training time: {:.2f}'.format(time.time() - start))
            start = time.time()
            te, n = None, 0
            for test_data in test_dataset:
                timg = test_data[0] if config.label else test_data
                t = gm.test_step_lg_gm_vae(model, timg)
                te = t.clone() if te is None else te + t
                n += 1
            print('Testing time: {:.2f}'.format(time.time() - start))
            tr = (acc / n_acc).tolist()
            te = (te / max(n, 1)).tolist() if te is not None else [float('nan')] * 6
            print('Training step {}'.format(step))
            for tag, v in (('', tr), ('Test ', te)):
                print('            {}X Recon Loss: {:.4f}, {}X KLD loss: {:.4f}, {}X hat Recon Loss: {:.4f}, {}X hat KLD loss: {:.4f}, '
                      '{}Y KL loss: {:.4f}'.format(tag, v[0], tag, v[1], tag, v[2], tag, v[3], tag, v[4]))
            _write_grids(model, test_dataset, config, os.path.join("output", RUN_NAME), step)
            acc, n_acc = None, 0
            start = time.time()
        if step >= config.training_steps:
            print('Training done!')
            break
    return _save(model, RUN_NAME)
