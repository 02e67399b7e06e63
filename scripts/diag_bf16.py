"""Diagnostic (GPU): per-tensor gradient error of the bf16 step vs the fp64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import np_ref, torch_ref
from split_vae_amd import ops
from split_vae_amd._lib import PHASE_ALL, PHASE_ADAM
from tests.test_gpu_step import make_inputs, flat_params, unflat, outputs10, NAMES10, LOSS_KEYS

for dtype in (torch.bfloat16, torch.float32):
    B, H, patch, beta = 8, 64, 8, 120.0
    x, perm, eps = make_inputs(B, H, patch, seed=5)
    images = ops.scramble_gather(torch.from_numpy(x).cuda(), torch.from_numpy(perm).cuda(), patch)
    params_np = np_ref.glorot_init(H, H, seed=3)
    ref = torch_ref.RefTrainer(params_np, beta, dtype=torch.float64)
    fwd_ref, loss_ref, g_ref = ref.grads(images.cpu().double(), eps[0], eps[1])
    plan = ops.LGVaePlan(B, H, H, beta=beta, dtype=dtype)
    P = flat_params(plan, params_np); G = torch.zeros_like(P)
    plan.step(PHASE_ALL & ~PHASE_ADAM, params=P, grads=G, images6=images, eps_x=torch.from_numpy(eps[0]).cuda(), eps_x_hat=torch.from_numpy(eps[1]).cuda())
    torch.cuda.synchronize()
    got = outputs10(plan, B, H)
    print("== dtype", dtype)
    for name, r in zip(NAMES10, fwd_ref):
        e = (got[name].cpu().double() - r.detach())
        print("  %-16s maxerr %.3e  relfro %.3e" % (name, float(e.abs().max()), float(e.norm() / r.norm())))
    losses = plan.buffer("losses", torch.float32, (8,)).cpu().double()
    for i, k in enumerate(LOSS_KEYS):
        print("  %-18s %.6f ref %.6f rel %.2e" % (k, float(losses[i]), float(loss_ref[k]), abs(float(losses[i]) - float(loss_ref[k])) / abs(float(loss_ref[k]))))
    for (name, off, shape), gr, gg in zip(plan.param_table, g_ref, unflat(plan, G)):
        gg = gg.double(); e = gg - gr
        print("  %-28s max|g| %.3e maxerr/max %.3e relfro %.3e cos %.6f" % (name, float(gr.abs().max()), float(e.abs().max() / gr.abs().max()), float(e.norm() / gr.norm()), float((gg * gr).sum() / (gg.norm() * gr.norm()))))
