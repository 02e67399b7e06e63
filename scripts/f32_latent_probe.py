"""Ad hoc: the x-hat encoder's incoming gradient at fp32, B images (SVHN-32): plan buffers against the fp64 oracle's intermediates (retain_grad)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from oracle import np_ref, torch_ref as R
from split_vae_amd import ops
from split_vae_amd._lib import PHASE_ALL, PHASE_ADAM
B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
H, patch, beta, L = (int(sys.argv[2]) if len(sys.argv) > 2 else 32), (int(sys.argv[3]) if len(sys.argv) > 3 else 1), 40.0, 128
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
plan = ops.LGVaePlan(B, H, H, beta=beta, dtype=torch.float32)
flat = torch.zeros(plan.n_params, dtype=torch.float32)
for (name, off, shape), p in zip(plan.param_table, params_np):
    flat[off:off + p.size] = torch.from_numpy(np.ascontiguousarray(p)).flatten()
P = flat.cuda(); G = torch.zeros_like(P)
plan.step(PHASE_ALL & ~PHASE_ADAM, params=P, grads=G, images6=images, eps_x=torch.from_numpy(eps_x).cuda(), eps_x_hat=torch.from_numpy(eps_h).cuda(), t=1)
torch.cuda.synchronize()

p = [torch.from_numpy(q).double().requires_grad_(True) for q in params_np]
im = images.cpu().double()
keep = {}
def enc(xx, pp, eps, tag):
    h = R.conv2d_same(xx, pp[0], pp[1], 2, 'relu'); h = R.conv2d_same(h, pp[2], pp[3], 2, 'relu'); h = R.conv2d_same(h, pp[4], pp[5], 2, 'relu')
    h.retain_grad(); keep["a3_" + tag] = h
    f = h.reshape(h.shape[0], -1)
    zm = f @ pp[6] + pp[7]; zs = F.softplus(f @ pp[8] + pp[9])
    zm.retain_grad(); zs.retain_grad(); keep["zm_" + tag] = zm; keep["zs_" + tag] = zs
    z = zm + zs * torch.from_numpy(eps).double()
    z.retain_grad(); keep["z_" + tag] = z
    return z, zm, zs
zx, zmx, zsx = enc(im[..., :3], p[0:10], eps_x, "x")
zh, zmh, zsh = enc(im[..., 3:], p[10:20], eps_h, "xh")
fwd = (*R.decoder(torch.cat([zx, zh], 1), p[20:30], H, H), zx, zmx, zsx, zh, *R.decoder(zh, p[30:40], H, H), zmh, zsh)
fwd = (fwd[0], fwd[1], zx, zmx, zsx, zh, fwd[6], fwd[7], zmh, zsh)
R.lgvae_losses(im, fwd, beta)["total_loss"].backward()
def cmp(tag, got, ref):
    d = (got - ref).abs(); m = float(ref.abs().max()) + 1e-300
    bad = d > 1e-4 * m
    rows = sorted(set(bad.nonzero()[:, 0].tolist()))[:16] if bad.any() else []
    print("  %-10s %.3e  bad %d  rows %s" % (tag, float(d.max()) / m, int(bad.sum()), rows))
for sfx in ("x", "xh"):
    cmp("ga3_" + sfx, plan.buffer("ga3_" + sfx, torch.float32, (B, H // 8, H // 8, 128)).cpu().double(), keep["a3_" + sfx].grad * (keep["a3_" + sfx] > 0))
    gh = plan.buffer("ghead_" + sfx, torch.float32, (B, 256)).cpu().double()
    cmp("ghead.m_" + sfx, gh[:, :128], keep["zm_" + sfx].grad)
gz_x = plan.buffer("gz_x", torch.float32, (B, 256)).cpu().double(); gz_xh = plan.buffer("gz_xh", torch.float32, (B, 128)).cpu().double()
print("  (dz of z_x = gz_x[:, :128]; dz of z_xh = gz_x[:, 128:] + gz_xh -- unless the slabs carry them: then gz_* are stale)")
cmp("dz_x", gz_x[:, :128], keep["z_x"].grad - 0)     # z.grad includes only the decoder path (KL acts on zm / zs)
cmp("dz_xh", gz_x[:, 128:] + gz_xh, keep["z_xh"].grad)

# ---- gate check of decoder x-hat: ReLU units whose gate (activation > 0) differs between the device step and the fp64 oracle, with the oracle's pre-activation there
print("== ReLU gates of the decoders (device activation > 0 vs fp64 pre-activation > 0)")
def dec_pre(z, pp):
    outs = []
    a = z @ pp[0] + pp[1]; outs.append(a.reshape(-1, H // 8, H // 8, 128)); h = F.relu(a).reshape(-1, H // 8, H // 8, 128)
    a = R.conv2d_same(h, pp[2], pp[3], 1, None); outs.append(a); h = F.relu(a)
    a = R.conv2d_same(R.resize_bilinear_2x(h), pp[4], pp[5], 1, None); outs.append(a); h = F.relu(a)
    a = R.conv2d_same(R.resize_bilinear_2x(h), pp[6], pp[7], 1, None); outs.append(a)
    return outs
with torch.no_grad():
    for tag, z, pp, sfx in (("x", torch.cat([zx, zh], 1), p[20:30], "x"), ("x_hat", zh, p[30:40], "xh")):
        pres = dec_pre(z.detach(), [q.detach() for q in pp])
        for name, pre, shp in zip(("h1_", "h2_", "h3_", "h4_"), pres, ((B, H // 8, H // 8, 128), (B, H // 8, H // 8, 128), (B, H // 4, H // 4, 64), (B, H // 2, H // 2, 32))):
            got = plan.buffer(name + sfx, torch.float32, shp).cpu().double()
            mism = (got > 0) != (pre > 0)
            idx = mism.nonzero()
            print("  decoder_%s %s: %d gate mismatches %s" % (tag, name, int(mism.sum()),
                  [(int(i[0]), "pre64 %.2e" % float(pre[tuple(i)]), "dev %.2e" % float(got[tuple(i)])) for i in idx[:6]]))

# ---- the encoders: ReLU gates against fp64 pre-activations, and each input gradient against an fp64 recomputation from the plan's own upstream buffers
print("== encoders: gates, then ga2 / ga1 recomputed in fp64 from the device's ga3 / ga2 (max |got - ref| / max |ref|, elements beyond 1e-4)")
with torch.no_grad():
    P64 = {name: torch.from_numpy(q).double() for (name, off, shape), q in zip(plan.param_table, params_np)}
for net, sfx, ch0 in (("encoder_x", "x", 0), ("encoder_x_hat", "xh", 3)):
    xin = im[..., ch0:ch0 + 3]
    shp = {"a1_": (B, H // 2, H // 2, 32), "a2_": (B, H // 4, H // 4, 64), "a3_": (B, H // 8, H // 8, 128)}
    dev = {k: plan.buffer(k + sfx, torch.float32, v).cpu().double() for k, v in shp.items()}
    pre1 = R.conv2d_same(xin, P64[net + "/e1/kernel"], P64[net + "/e1/bias"], 2, None)
    pre2 = R.conv2d_same(F.relu(pre1), P64[net + "/e2/kernel"], P64[net + "/e2/bias"], 2, None)
    pre3 = R.conv2d_same(F.relu(pre2), P64[net + "/e3/kernel"], P64[net + "/e3/bias"], 2, None)
    for k, pre in (("a1_", pre1), ("a2_", pre2), ("a3_", pre3)):
        mism = (dev[k] > 0) != (pre > 0)
        idx = mism.nonzero()
        print("  %s %s: %d gate mismatches %s" % (net, k, int(mism.sum()), [(int(i[0]), "pre64 %.2e" % float(pre[tuple(i)]), "dev %.2e" % float(dev[k][tuple(i)])) for i in idx[:4]]))
    ga3 = plan.buffer("ga3_" + sfx, torch.float32, shp["a3_"]).cpu().double()
    a2v = dev["a2_"].clone().requires_grad_(True)
    (g2,) = torch.autograd.grad((R.conv2d_same(a2v, P64[net + "/e3/kernel"], P64[net + "/e3/bias"], 2, None) * ga3).sum(), a2v)
    ga2 = plan.buffer("ga2_" + sfx, torch.float32, shp["a2_"]).cpu().double(); cmp(sfx + " ga2", ga2, g2 * (dev["a2_"] > 0))
    a1v = dev["a1_"].clone().requires_grad_(True)
    (g1,) = torch.autograd.grad((R.conv2d_same(a1v, P64[net + "/e2/kernel"], P64[net + "/e2/bias"], 2, None) * ga2).sum(), a1v)
    ga1 = plan.buffer("ga1_" + sfx, torch.float32, shp["a1_"]).cpu().double(); cmp(sfx + " ga1", ga1, g1 * (dev["a1_"] > 0))
