"""GPU: caller-owned workspaces need no particular contents (include/splitvae.h).

The C ABI takes raw device blocks; a C caller's hipMalloc'd memory is garbage.  Round 5 found the plan reading workspace it had never written (pad channels, accumulators):
with 0xFF bytes in the block every loss was NaN, while the Python mirror -- which allocated zeros -- never noticed.  sv_lgvae_plan_bind / sv_gm_encoder_bind zero-fill the
block now; the per-operator workspaces (weight-gradient slabs, the polyphase input gradient's edge terms) are written before they are read."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plan_workspace_may_hold_anything(lib_built):
    """Seven SPLIT-VAE configurations (fp32 / bf16, 5 ... 512 images) and SPLIT-GMVAE at both precisions, three training steps each: losses, gradients and weights hash
    identically whether the blocks held zeros or 0xFF bytes before sv_lgvae_plan_bind / sv_gm_encoder_bind (scripts/ws_poison_probe.py)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "ws_poison_probe.py")], capture_output=True, text=True, cwd=ROOT, timeout=900)
    lines = [l for l in r.stdout.splitlines() if " zeros " in l]
    assert len(lines) == 9, r.stdout + r.stderr[-1500:]
    assert r.returncode == 0 and all(l.rstrip().endswith("SAME") for l in lines), "\n".join(lines)


@pytest.mark.parametrize("layer", [("d4", 32, 64, 32, 6, False), ("d5", 64, 32, 6, 6, True), ("e2", 32, 32, 64, 6, False)], ids=lambda l: l[0])
def test_operator_workspaces_may_hold_anything(lib_built, layer):
    """sv_conv2d_nhwc_wgrad_ws / sv_conv2d_nhwc_dgrad_lowres_ws at fp32 with the caller's block filled with 0xFF bytes first: bitwise the result of a zeroed block."""
    from split_vae_amd import ops
    name, H, Cin, Cout, k, yf32 = layer
    ups = name.startswith("d")
    s = 1 if ups else 2
    B = 6
    rng = np.random.default_rng(11)
    c8 = (Cout + 7) // 8 * 8
    w = torch.from_numpy(rng.uniform(-1, 1, (k, k, Cin, Cout)).astype(np.float32)) * math.sqrt(6.0 / (k * k * (Cin + Cout)))
    conv = ops.Conv2D(B, H, H, Cin, Cout, k, s, act=None, dtype=torch.float32, y_f32=yf32, ups_in=ups)
    conv.prep(w.cuda())
    hin = H // 2 if ups else H
    x = torch.from_numpy(rng.standard_normal((B, hin, hin, conv.desc.ldx)).astype(np.float32)).cuda()
    dy = torch.from_numpy(rng.standard_normal((B, H // s, H // s, c8)).astype(np.float32))
    dy[..., Cout:] = 0
    dy = dy.cuda()
    out = {}
    for fill in (0, 255):
        conv._ws = None
        conv._dws = None
        dw, db = conv.wgrad(x, dy, workspace=True)            # allocates conv._ws
        conv._ws.fill_(fill)
        dw, db = conv.wgrad(x, dy, workspace=True)
        res = [dw.clone(), db.clone()]
        if ups:
            dx = conv.dgrad_lowres(dy)                        # allocates conv._dws
            if dx is not None and getattr(conv, "_dws", None) is not None:
                conv._dws.fill_(fill)
                res.append(conv.dgrad_lowres(dy).clone())
        torch.cuda.synchronize()
        out[fill] = res
    for a, b in zip(out[0], out[255]):
        assert torch.equal(a, b)
        assert not torch.isnan(b).any()
