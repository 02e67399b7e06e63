"""Probe of tests/test_gpu_dist.py::test_buckets_wait_for_the_side_stream: one step with SV_PHASE_BUCKET_EVENTS, then, on a fresh stream ordered only by
sv_lgvae_bucket_wait, a snapshot of every bucket's gradient range.  Prints the relative distance of each snapshot from the final gradients
(0 = the stream really waited for the bucket, side-stream work included).  usage: python tests/bucket_probe.py  (knobs in the environment)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import split_vae_amd                                                 # noqa: E402
split_vae_amd.configure_hw_queues()                                  # the library's queue setting, before the first HIP call
from split_vae_amd import _lib, data, dist as svdist                 # noqa: E402
from split_vae_amd.augmentation import Augmentator                   # noqa: E402
from split_vae_amd.model import LGVae                                # noqa: E402
from split_vae_amd.optimizer import Adam                             # noqa: E402


def main():
    B, H = 16, 32
    model = LGVae(128, 128, image_shape=[-1, H, H, 3], dtype="f32", device="cuda", seed=3)
    model.beta = 40.0
    opt = Adam(learning_rate=1e-3)
    x = data.synthetic_images(B, H, H, seed=0, device="cuda")
    img = Augmentator("scramble", size=4, seed=1).augment(x)
    plan = model.plan(B)
    # the plan's own test hooks (include/splitvae.h: sv_lgvae_plan_debug); the library itself reads no environment variable for them
    plan.debug("side_delay_us", int(os.environ.get("SV_PYTEST_SIDE_DELAY_US", "0")))
    plan.debug("bucket_skip_side", int(os.environ.get("SV_PYTEST_BUCKET_SKIP_SIDE", "0")))
    m, v = opt.slots(model.flat)
    buckets = svdist.param_buckets(model.param_table, model.n_params)
    names = {0: "decoders", 1: "enc_heads", 2: "enc_convs"}
    out = []
    for it in range(2):                                               # the second step is the measured one (lazy allocations, LDS caps out of the way)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        plan.step((_lib.PHASE_ALL & ~_lib.PHASE_ADAM) | _lib.PHASE_BUCKET_EVENTS, params=model.flat, grads=model.grad_flat, adam_m=m, adam_v=v,
                  images6=img, seed=3, step=it)
        snaps = {}
        for k, name in names.items():
            s = torch.cuda.Stream()
            plan.bucket_wait(k, s)
            with torch.cuda.stream(s):
                snaps[k] = [model.grad_flat[b:e].clone() for b, e in buckets[name]]
                ek = torch.cuda.Event(enable_timing=True); ek.record(s)
                snaps[(k, "ev")] = ek
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        torch.cuda.synchronize()
        print("TIMES main_end %.2f ms, snapshots at" % e0.elapsed_time(e1), ["%.2f" % e0.elapsed_time(snaps[(k, "ev")]) for k in names])
        out = []
        for k, name in names.items():
            fin = torch.cat([model.grad_flat[b:e] for b, e in buckets[name]])
            snap = torch.cat(snaps[k])
            out.append(float((fin - snap).norm() / fin.norm().clamp_min(1e-30)))
    print("BUCKET_SNAPSHOT_ERR", *out)


if __name__ == "__main__":
    main()
