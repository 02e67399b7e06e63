"""Host-side mirror of augmentation.py:12-57 (Augmentator) for the SPLIT-VAE path.

Only `scramble` is on the path (vae/main.py:25 default; every README command uses it); the
index gather runs in the HIP kernel sv_scramble_gather, batched on the device instead of
per image inside tf.data (vae/main.py:57-61)."""
import torch

from . import ops


class Augmentator(object):
    def __init__(self, type, size=1, mean=0, std=1, seed=0):
        self.size = size
        self.seed = seed
        self._step = 0
        if type == 'scramble':
            self.augment = self.scramble
        elif type == 'no_op':
            self.augment = self.no_op
        elif type in ('mix_scramble', 'blur', 'high_low_pass'):
            # augmentation.py:59-101: never selected by any README command / config
            raise NotImplementedError("augmentation '%s' is outside the SPLIT-VAE hot path" % type)
        else:
            raise ValueError("unknown augmentation type %r" % (type,))

    def scramble(self, x, perm=None, sample_offset=0, plan=None):
        """x[B,H,W,3] (or [H,W,3]) fp32 on the device -> concat([x, x_aug], axis=-1).
        perm[B,(H/size)^2] int32 makes the shuffle explicit (the reference draws it from TF's
        unseeded RNG, augmentation.py:49); by default it comes from the counter-based Philox
        stream keyed by (seed, call index, global sample index)."""
        single = x.dim() == 3
        if single:
            x = x[None]
            if perm is not None:
                perm = perm[None]
        B, H, W, C = x.shape
        if H != W or H % self.size:
            raise ValueError("scramble assumes square images and size | H (augmentation.py:44-46)")
        if perm is None:
            perm = ops.random_perm(B, (H // self.size) * (W // self.size), self.seed, self._step, sample_offset, x.device)
            self._step += 1
        staged = None
        if plan is not None and not single and (plan.desc.B, plan.desc.H, plan.desc.W) == (B, H, W):
            # the training plan's padded input buffers are written by the same kernel (ops.scramble_gather, staged=): train_step
            # recognises the returned tensor and skips its split / pad pass
            staged = (plan.buffer("in8_x", plan.dtype, (B, H, W, 8)), plan.buffer("in8_xh", plan.dtype, (B, H, W, 8)))
        out = ops.scramble_gather(x.contiguous(), perm.to(torch.int32).contiguous(), self.size, staged=staged)
        if staged is not None:
            # valid for ONE train_step on this plan, and only while nothing else has written the plan's input buffers since
            # (another staged batch, a test_step / encode / visualizer call of the same batch size) and `out` is not edited in place:
            # trainer.train_step checks the generation and the tensor's version counter and falls back to its own split / pad pass
            plan.in8_gen += 1
            out._sv_staged_plan, out._sv_staged_gen, out._sv_staged_version = plan, plan.in8_gen, out._version
        return out[0] if single else out

    def no_op(self, x):
        return x
