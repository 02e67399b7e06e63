"""Ad hoc: at B = 96 SVHN-32 fp32, is the e1 kernel gradient's deviation from the fp64 oracle a property of fp32 (the CPU fp32 restatement shows the same) or of the HIP path?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import np_ref, torch_ref
from split_vae_amd import ops
from split_vae_amd._lib import PHASE_ALL, PHASE_ADAM
B, H, patch, beta, L = int(sys.argv[1]) if len(sys.argv) > 1 else 96, int(sys.argv[2]) if len(sys.argv) > 2 else 32, int(sys.argv[3]) if len(sys.argv) > 3 else 1, 40.0, 128
rng0 = np.random.Generator(np.random.PCG64(5))
x = (rng0.integers(0, 256, size=(B, H, H, 3)) / 255.0 * 2 - 1).astype(np.float32)
perm = np.stack([np.random.Generator(np.random.PCG64(6 + b)).permutation((H // patch) ** 2) for b in range(B)]).astype(np.int32)
eps_x = np.random.Generator(np.random.PCG64(7)).standard_normal((B, L)).astype(np.float32)
eps_h = np.random.Generator(np.random.PCG64(8)).standard_normal((B, L)).astype(np.float32)
images = ops.scramble_gather(torch.from_numpy(x).cuda(), torch.from_numpy(perm).cuda(), patch)
params_np = np_ref.glorot_init(H, H, seed=3)
rng = np.random.default_rng(9)
for i in range(1, len(params_np), 2):
    params_np[i] = (rng.standard_normal(params_np[i].shape) * 0.05).astype(np.float32)
r64 = torch_ref.RefTrainer(params_np, beta, dtype=torch.float64)
r32 = torch_ref.RefTrainer(params_np, beta, dtype=torch.float32)
_, _, g64 = r64.grads(images.cpu().double(), eps_x, eps_h)
_, _, g32 = r32.grads(images.cpu().float(), eps_x, eps_h)
plan = ops.LGVaePlan(B, H, H, beta=beta, dtype=torch.float32)
flat = torch.zeros(plan.n_params, dtype=torch.float32)
for (name, off, shape), p in zip(plan.param_table, params_np):
    flat[off:off + p.size] = torch.from_numpy(np.ascontiguousarray(p)).flatten()
P = flat.cuda(); G = torch.zeros_like(P)
plan.step(PHASE_ALL & ~PHASE_ADAM, params=P, grads=G, images6=images, eps_x=torch.from_numpy(eps_x).cuda(), eps_x_hat=torch.from_numpy(eps_h).cuda(), t=1)
torch.cuda.synchronize()
Gc = G.cpu()
print("B %d H %d: %-34s %12s %12s %12s" % (B, H, "variable", "hip-vs-f64", "cpuf32-vs-f64", "hip-vs-cpuf32"))
for (name, off, shape), a, b in zip(plan.param_table, g64, g32):
    n = int(np.prod(shape))
    gh = Gc[off:off + n].view(*shape).double()
    m = float(a.abs().max()) + 1e-30
    e1, e2, e3 = float((gh - a).abs().max()) / m, float((b.double() - a).abs().max()) / m, float((gh - b.double()).abs().max()) / m
    worst = max(e1, locals().get("worst", 0.0)) if False else None
    if max(e1, e2) > 5e-4:
        print("%-34s %12.2e %12.2e %12.2e" % (name, e1, e2, e3))
