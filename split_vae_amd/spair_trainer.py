"""SPAIR / SPLIT-SPAIR losses, step closures and loop (spair/trainer.py of 51616/split-vae) on the device operators.

`train_step(model, images, optimizer, step, config)` = spair/trainer.py:136-234: forward (training=True), the loss assembly of
the configured model, the gradient of every variable (torch autograd over the `split_vae::*` operators, which pair each forward
kernel with its hand-written adjoint), and tf.keras.optimizers.Adam(lr, clipnorm=1.0) (spair/main.py:109) as ONE pair of launches
over the model's flat variable buffer (sv_adam_step_clipnorm).  The sequential 16-cell count-prior KL of z_pres is one kernel
(spair_render.hip: sv_spair_zpres_kl); the Gaussian KLs and the cross-entropy are torch elementwise reductions.
"""
import time

import torch

from . import ops, torch_ops as T

TRAIN_METRIC_NAMES = ['x_recon_train_loss', 'z_zoom_kl_train_loss', 'z_what_kl_train_loss', 'z_where_kl_train_loss',
                      'z_depth_kl_train_loss', 'z_pres_kl_train_loss', 'z_bg_kl_train_loss', 'z_l_kl_train_loss',
                      'x_hat_recon_train_loss']                                                    # spair/trainer.py:125-126
TEST_METRIC_NAMES = [n.replace('_train_', '_test_') for n in TRAIN_METRIC_NAMES] + ['MAE test', 'MAPE test']   # :128-129


# The three reductions below run as one kernel each with their gradients (spair_loss.hip: sv_spair_loss); the elementwise
# restatements next to them (tf_safe_log, xent_loss) keep the reference's names for callers that want the per-element maps.
def tf_safe_log(value, replacement_value=-100.0):
    """spair/trainer.py:97-101."""
    lv = torch.log(value + 1e-8)
    return torch.where(torch.isnan(lv) | torch.isinf(lv), torch.full_like(lv, replacement_value), lv)


def tf_mean_sum(t):
    """:107-109: average over the batch, sum over everything else."""
    return t.reshape(t.shape[0], -1).sum(dim=1).mean()


def xent_loss(label, pred):
    """:103-104."""
    return -(label * tf_safe_log(pred) + (1.0 - label) * tf_safe_log(1.0 - pred))


def kl_divergence(z_mean, z_sig):
    """:13-21 (rank 2 and rank 4 inputs: sum over all but the batch axis)."""
    if z_mean.dim() not in (2, 4):
        raise NotImplementedError('This KL shape is not implemented')
    return T.spair_kl(z_mean.contiguous(), z_sig.contiguous()).mean()


def kl_divergence_two_gauss(mean1, sig1, mean2, sig2):
    """:23-24.  mean2: a python scalar or a 1-element device tensor, sig2: a python scalar (the constant zoom prior of
    :156-157: one kernel); full tensors take the composed expression."""
    if (not torch.is_tensor(mean2) or mean2.numel() == 1) and not torch.is_tensor(sig2):
        return T.spair_kl_prior(mean1.contiguous(), sig1.contiguous(), mean2, sig2).mean()
    return tf_mean_sum(tf_safe_log(sig2) - tf_safe_log(sig1) + (sig1 * sig1 + (mean1 - mean2) ** 2) / (2 * sig2 * sig2) - 0.5)


def xent_mean_sum(label, pred):
    """tf_mean_sum(xent_loss(label, pred)) (:148, :186) as one kernel."""
    return T.spair_xent(label.contiguous(), pred.contiguous()).mean()


def compute_z_pres_kl_yolo_air(z_pres, z_pres_logits, z_pres_pre_sigmoid, prior_prob, temperature):
    """:45-94 as one kernel (per-image sums) + the batch mean."""
    return T.spair_zpres_kl(z_pres.contiguous(), z_pres_logits.contiguous(), z_pres_pre_sigmoid.contiguous(), prior_prob, temperature).mean()


class ClipnormAdam:
    """tf.keras.optimizers.Adam(learning_rate, clipnorm=...) (spair/main.py:109) over a model's flat variable buffer.

    [TF-2.0 semantics] The reference's step is tape.gradient -> optimizer.apply_gradients (spair/trainer.py:226-227).  Under the
    pinned tensorflow_gpu==2.0.0 (requirements.txt:7) OptimizerV2 applies `clipnorm` only inside get_gradients /
    _compute_gradients (the minimize / fit paths); apply_gradients does not clip (it does from TF 2.4 on).  So with the pinned
    version Adam(lr, clipnorm=1.0) is a plain Keras Adam in this training loop: `clip_in_apply=False`, the default, like
    --gm_dropout's default follows the pinned version; `clip_in_apply=True` (--clipnorm_semantics tf2.4) clips each gradient
    tensor with tf.clip_by_norm before the update.  Not executable here (no TensorFlow): documented switch, both tested."""

    def __init__(self, learning_rate=1e-4, clipnorm=1.0, beta_1=0.9, beta_2=0.999, epsilon=1e-7, clip_in_apply=False):
        self.learning_rate = learning_rate
        self.clip_in_apply = bool(clip_in_apply)
        # clipnorm >= 1e37 selects the kernels' plain (unclipped) Keras-Adam update: the scale is exactly the gradient scale, whatever the
        # norm is (as the ratio clipnorm / max(||g||, clipnorm) an infinite norm zeroed the tensor instead of passing through)
        self.clipnorm = clipnorm if self.clip_in_apply else 3.0e38
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon
        self.iterations = 0
        self._slots = None

    def alpha(self, t):
        return ops.adam_alpha(self.learning_rate, self.beta_1, self.beta_2, t)

    def slots(self, flat):
        """Adam's m / v for the model's flat variable buffer."""
        if self._slots is None:
            self._slots = (torch.zeros_like(flat), torch.zeros_like(flat))
        return self._slots

    def apply_gradients(self, model, grads, alpha_dev=None):
        """alpha_dev: 1-element device tensor with alpha(iterations + 1) -- the captured (hipGraph) step reads the step size from it
        and the caller advances `iterations` per replay."""
        st = model.store
        if self._slots is None:
            self._slots = (torch.zeros_like(st.flat), torch.zeros_like(st.flat))
        m, v = self._slots
        if alpha_dev is None:
            self.iterations += 1
        gs = [g if (g.is_contiguous() and g.data_ptr() % 16 == 0) else g.contiguous().clone() for g in grads]
        # variables start 16-B aligned in the flat buffer: a gradient whose size is not a multiple of 4 is zero-padded to its slot
        gs = [g if g.numel() % 4 == 0 else torch.nn.functional.pad(g.reshape(-1), (0, (-g.numel()) % 4)) for g in gs]
        if len(gs) <= 128:                                # the gradient tensors as they are: their addresses go to the kernels by value
            ops.adam_step_clipnorm_tensors(st.flat, gs, m, v, st.tensor_off, self.clipnorm, max(self.iterations, 1), float(self.learning_rate),
                                           self.beta_1, self.beta_2, self.epsilon, alpha_dev=alpha_dev)
        else:
            ops.adam_step_clipnorm(st.flat, torch.cat([x.reshape(-1) for x in gs]), m, v, st.tensor_off, self.clipnorm, max(self.iterations, 1),
                                   float(self.learning_rate), self.beta_1, self.beta_2, self.epsilon, alpha_dev=alpha_dev)


def _unpack(config, out):
    names = ["x_recon", "z_what", "z_what_mean", "z_what_sigma", "z_where", "z_where_mean", "z_where_sigma", "z_depth", "z_depth_mean",
             "z_depth_sigma", "z_pres", "z_pres_logits", "z_pres_pre_sigmoid", "all_glimpses", "obj_recon_unnorm", "obj_recon_alpha",
             "obj_full_recon_unnorm", "obj_bbox_mask"]
    if config.model == "lg_spair":
        names += ["z_bg", "z_bg_mean", "z_bg_sig", "x_hat_recon", "z_l", "z_l_mean", "z_l_sig"]
    elif config.model == "bg_spair":
        names += ["z_bg", "z_bg_mean", "z_bg_sig"]
    return dict(zip(names, out))


def step_scalars(config, step, training=True):
    """The step-dependent constants of train_step (:153, :156, :165-167): prior_z_pres_prob, the zoom prior's mean and the
    annealed beta of the spair / bg_spair objectives; test_step (:247-250) uses their final values."""
    anneal = min(1.0, (step + 1) / config.z_pres_anneal_step) if training else 1.0
    return {"prior_prob": 0.99 * anneal,
            "zoom_mean": config.prior_z_zoom + (config.prior_z_zoom_start * (1 - anneal) if training else 0.0),
            "annealed_beta": min(config.beta, config.beta * (step + 1.0) / config.anneal_until)}


def compute_losses(config, images, out, step, training=True, dyn=None):
    """The loss assembly of train_step (:142-228; training=True) or test_step (:243-293) -> (total_loss | None, [losses]).
    dyn: step_scalars as 1-element device tensors (the captured step reads them at replay time) instead of `step`."""
    o = _unpack(config, out)
    lg = config.model == "lg_spair"
    sc = dyn if dyn is not None else step_scalars(config, step, training)
    x = images[..., :3]                                               # :148-151 (spair / bg_spair canvases have 3 channels)
    x_recon_loss = xent_mean_sum(x, o["x_recon"])
    z_pres_kl = compute_z_pres_kl_yolo_air(o["z_pres"], o["z_pres_logits"], o["z_pres_pre_sigmoid"], sc["prior_prob"], config.tau)
    zm, zs = o["z_where_mean"], o["z_where_sigma"]
    zoom_kl = kl_divergence_two_gauss(zm[..., :2], zs[..., :2], sc["zoom_mean"], 0.5)
    what_kl = kl_divergence(o["z_what_mean"], o["z_what_sigma"])
    where_kl = kl_divergence(zm[..., 2:], zs[..., 2:])
    depth_kl = kl_divergence(o["z_depth_mean"], o["z_depth_sigma"])
    losses = [x_recon_loss, zoom_kl, what_kl, where_kl, depth_kl, z_pres_kl]
    if not training:                                                                               # test_step :262-285
        if lg:
            losses += [kl_divergence(torch.cat([o["z_bg_mean"], o["z_l_mean"]], dim=1), torch.cat([o["z_bg_sig"], o["z_l_sig"]], dim=1)),
                       kl_divergence(o["z_l_mean"], o["z_l_sig"]), xent_mean_sum(images[..., 3:], o["x_hat_recon"])]
        elif config.model == "bg_spair":
            losses += [kl_divergence(o["z_bg_mean"], o["z_bg_sig"]), torch.zeros((), device=x.device), torch.zeros((), device=x.device)]
        return None, losses
    rw = config.reconstruction_weight
    obj = lambda wk: config.z_what_beta * wk + depth_kl + where_kl + zoom_kl + z_pres_kl
    annealed_beta = sc["annealed_beta"]
    if torch.is_tensor(annealed_beta):
        annealed_beta = annealed_beta.reshape(())
    if lg:
        x_hat_recon_loss = xent_mean_sum(images[..., 3:], o["x_hat_recon"])
        z_l_kl = kl_divergence(o["z_l_mean"], o["z_l_sig"])
        if not config.split_z_l:                                                                   # :176-195
            if config.concat_z_bg:
                z_bg_kl = kl_divergence(torch.cat([o["z_bg_mean"], o["z_l_mean"]], dim=1), torch.cat([o["z_bg_sig"], o["z_l_sig"]], dim=1))
            else:
                z_bg_kl = kl_divergence(o["z_bg_mean"], o["z_bg_sig"])
            if config.concat_z_what:
                tile = lambda t: t[:, None, None, :].expand(-1, 4, 4, -1)
                what_kl = kl_divergence(torch.cat([o["z_what_mean"], tile(o["z_l_mean"])], dim=-1),
                                        torch.cat([o["z_what_sigma"], tile(o["z_l_sig"])], dim=-1))
            total = config.z_bg_beta * z_bg_kl + rw * x_recon_loss + config.beta * obj(what_kl) + x_hat_recon_loss
        else:                                                                                      # :197-207
            z_bg_kl = kl_divergence(o["z_bg_mean"], o["z_bg_sig"])
            total = config.z_bg_beta * z_bg_kl + config.z_l_beta * z_l_kl + x_hat_recon_loss + rw * x_recon_loss + config.beta * obj(what_kl)
        losses += [z_bg_kl, z_l_kl, x_hat_recon_loss]
    elif config.model == "bg_spair":                                                               # :222-228
        z_bg_kl = kl_divergence(o["z_bg_mean"], o["z_bg_sig"])
        losses.append(z_bg_kl)
        total = config.z_bg_beta * z_bg_kl + rw * x_recon_loss + annealed_beta * obj(what_kl)
    else:                                                                                          # :165-167
        total = rw * x_recon_loss + annealed_beta * obj(what_kl)
    return total, losses


_TAPE_MAX = 16          # sv_tape_loss_info: [total, reported x 16, mean loss_i x 16]

RETURN_NAMES = ["x_recon", "z_what", "z_what_mean", "z_what_sigma", "z_where", "z_where_mean", "z_where_sigma", "z_depth", "z_depth_mean",
                "z_depth_sigma", "z_pres", "z_pres_logits", "z_pres_pre_sigmoid", "all_glimpses", "obj_recon_unnorm", "obj_recon_alpha",
                "obj_full_recon_unnorm"]


def _native_ok(model):
    """The native launch sequence (spair_native.NativeStep) is the step; SV_SPAIR_AUTOGRAD=1 keeps the torch-autograd graph over the
    split_vae::* operators (round 2's form: A/B and the operator-level tests)."""
    import os
    return not os.environ.get("SV_SPAIR_AUTOGRAD")


def _return_names(config):
    extra = ["z_bg", "z_bg_mean", "z_bg_sig", "x_hat_recon", "z_l", "z_l_mean", "z_l_sig"] if config.model == "lg_spair" else \
        (["z_bg", "z_bg_mean", "z_bg_sig"] if config.model == "bg_spair" else [])
    return RETURN_NAMES + extra                                          # the step's return drops obj_bbox_mask (:230-232)


def train_step_native(model, images, optimizer, step, config, noise=None, return_grads=False, accumulate_metrics=False):
    """train_step as ONE native call (sv_tape_run): forward, loss assembly, adjoint, Adam -- no autograd engine, no ATen kernels."""
    ns = model.native(images.shape[0], config, training=True)
    lo = ns.run(images, step_scalars(config, float(step), True), optimizer=optimizer, noise=noise, backward=True,
                accumulate_metrics=accumulate_metrics)
    res = ns.outputs(_return_names(config))
    losses = ns._loss_list                                                # views of the tape's loss block (overwritten by the next step)
    return (res, losses, lo[0], ns.grad_views()) if return_grads else (res, losses)


def train_step(model, images, optimizer, step, config, noise=None, return_grads=False):
    """spair/trainer.py:136-234.  `noise`: pinned random draws by name (tests); default = the model's device generator."""
    if _native_ok(model):
        return train_step_native(model, images, optimizer, step, config, noise, return_grads)
    out = model(images, training=True, noise=noise)
    total_loss, losses = compute_losses(config, images, out, float(step), training=True)
    variables = [v for _, v in model.trainable_variables]
    grads = torch.autograd.grad(total_loss, variables, allow_unused=True)
    grads = [g if g is not None else torch.zeros_like(v) for g, v in zip(grads, variables)]
    optimizer.apply_gradients(model, grads)
    o = tuple(t.detach() if torch.is_tensor(t) else t for t in out)
    res = o[:17] + o[18:]                                              # the reference drops obj_bbox_mask from the step's return (:230-232)
    return (res, [l.detach() for l in losses], total_loss.detach(), grads) if return_grads else (res, [l.detach() for l in losses])


class GraphedTrainStep:
    """train_step captured once into a hipGraph (torch.cuda.graph over the same operators) and replayed: the ~700 launches of a
    step cost one graph launch on the host.  The step-dependent scalars (step_scalars, Adam's step size) live in a 4-float
    device buffer refreshed before every replay; the batch is copied into a static buffer; the random draws come from torch's
    default device generator (graph-safe).  Returns the static output tensors of the captured step (overwritten by the next
    replay: clone what must be kept)."""

    def __init__(self, model, optimizer, config, images_like, warmup=3, noise=None):
        self.model, self.optimizer, self.config, self.noise = model, optimizer, config, noise      # noise: pinned draws (tests)
        self.images = torch.empty_like(images_like)
        self.images.copy_(images_like)
        self.dyn = torch.zeros((4,), dtype=torch.float32, device=images_like.device)
        self._views = {"prior_prob": self.dyn[0:1], "zoom_mean": self.dyn[1:2], "annealed_beta": self.dyn[2:3]}
        model.generator = None                                            # default generator: its Philox offset advances per replay
        self._stage(0)
        st = model.store
        saved = st.flat.clone()                                           # the warm-up steps are real steps: undone below
        saved_mv = [t.clone() for t in optimizer._slots[:2]] if optimizer._slots is not None else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                                     # allocator / lazy-init warm-up off the capture stream
            for _ in range(warmup):
                self._body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        st.flat.copy_(saved)
        for i, t in enumerate(optimizer._slots[:2]):
            t.copy_(saved_mv[i]) if saved_mv is not None else t.zero_()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs, self.losses, self.total = self._body()

    def _stage(self, step):
        sc = step_scalars(self.config, float(step), True)
        self.dyn.copy_(torch.tensor([sc["prior_prob"], sc["zoom_mean"], sc["annealed_beta"],
                                     self.optimizer.alpha(self.optimizer.iterations + 1)], dtype=torch.float32), non_blocking=True)

    def _body(self):
        out = self.model(self.images, training=True, noise=self.noise)
        total, losses = compute_losses(self.config, self.images, out, 0.0, training=True, dyn=self._views)
        variables = [v for _, v in self.model.trainable_variables]
        grads = torch.autograd.grad(total, variables, allow_unused=True)
        grads = [g if g is not None else torch.zeros_like(v) for g, v in zip(grads, variables)]
        self.optimizer.apply_gradients(self.model, grads, alpha_dev=self.dyn[3:4])
        o = tuple(t.detach() for t in out)
        return o[:17] + o[18:], [l.detach() for l in losses], total.detach()

    def __call__(self, images, step):
        self.images.copy_(images, non_blocking=True)
        self._stage(step)
        self.graph.replay()
        self.optimizer.iterations += 1
        return self.outputs, self.losses


@torch.no_grad()
def test_step(model, images, config, labels=None, noise=None):
    """spair/trainer.py:236-308 (the reference evaluates with model(images, training=True) too)."""
    if _native_ok(model) and labels is None:
        ns = model.native(images.shape[0], config, training=True)
        lo = ns.run(images, step_scalars(config, 0.0, False), noise=noise, backward=False)
        names = _return_names(config)
        # mean loss_i of the tape: [x_recon, zoom, what, where, depth, z_pres, (bg, l, x_hat)]
        means = [lo[1 + _TAPE_MAX + i] for i in range(ns.n_loss)]
        losses = means[:6]
        if config.model == "lg_spair":                                  # test_step :262-285
            losses = losses + [means[6] + means[7], means[7], means[8]]
        elif config.model == "bg_spair":
            z = torch.zeros((), device=images.device)
            losses = losses + [means[6], z, z]
        return ns.outputs(names), losses
    out = model(images, training=True, noise=noise)
    _, losses = compute_losses(config, images, out, 0.0, training=False)
    if labels is not None:
        pred_count = torch.round(torch.sigmoid(out[11])).sum(dim=(1, 2, 3))
        lab = labels.to(pred_count)
        losses.append((lab - pred_count).abs().mean())
        losses.append(100.0 * ((lab - pred_count).abs() / lab.abs().clamp_min(1e-7)).mean())
    return out[:17] + out[18:], losses


def train_spair(model, optimizer, dataset, train_dataset, test_dataset, config, log=print):
    """spair/trainer.py:112-424 without the matplotlib grids: the step loop, the 1000-step metric logs, save_weights at the end."""
    sums, n = None, 0
    start = time.time()
    history = []
    every = int(config.log_every or 1000)
    graphed = None
    for step, images in enumerate(train_dataset):
        if config.graph:                                                  # one hipGraph replay per step (GraphedTrainStep)
            if graphed is None:
                graphed = GraphedTrainStep(model, optimizer, config, images)
            _, losses = graphed(images, step)
        else:
            _, losses = train_step(model, images, optimizer, step, config)
        vals = torch.stack([l.float() for l in losses])
        sums = vals if sums is None else sums + vals
        n += 1
        if step % every == 0:
            torch.cuda.synchronize()
            log('Training time: {:.2f}'.format(time.time() - start))
            tr = dict(zip(TRAIN_METRIC_NAMES, (sums / n).tolist()))
            log('Training step:', step)
            log(tr)
            sums, n = None, 0
            rec = {"step": step, "train": tr}
            for test_num, test_ds in enumerate(test_dataset or []):
                ts, tn = None, 0
                for batch in test_ds:
                    imgs, labels = batch if isinstance(batch, (tuple, list)) else (batch, None)
                    _, tl = test_step(model, imgs, config, labels)
                    tv = torch.stack([l.float() for l in tl])
                    ts = tv if ts is None else ts + tv
                    tn += 1
                if tn:
                    te = dict(zip([nm + str(test_num) for nm in TEST_METRIC_NAMES], (ts / tn).tolist()))
                    log(te)
                    rec["test" + str(test_num)] = te
            history.append(rec)
            start = time.time()
        if step >= config.training_steps:
            log('Training done!')
            break
    if config.save_weights:                                               # spair/trainer.py:424 (the reference always saves; opt-in here)
        import os
        from . import h5io
        os.makedirs('models', exist_ok=True)
        log('saved', model.save_weights(os.path.join('models', time.strftime("%Y%m%d-%H%M%S") + ('.h5' if h5io.available() else '.npz'))))
    return history
