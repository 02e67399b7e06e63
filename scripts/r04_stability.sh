cd $GRAFT_REPO_ROOT
echo "== long run bf16 B=256, 3000 steps"; timeout 600 python scripts/long_run.py 3000 2>&1 | grep -v amdgpu | tail -8
echo "== step hash twice (run-to-run identical bf16 steps at B=64 / B=512)"; for i in 1 2; do timeout 300 python scripts/r03_step_hash.py 2>&1 | grep -v amdgpu | tail -3; done
echo "== fp32 step hash twice (B=512, default mode: slab weight gradients)"; for i in 1 2; do timeout 300 python - <<'PY'
import hashlib, os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from split_vae_amd import data, trainer
from split_vae_amd.augmentation import Augmentator
from split_vae_amd.model import LGVae
from split_vae_amd.optimizer import Adam
x = data.synthetic_images(512, 64, 64, seed=0, device="cuda")
img = Augmentator("scramble", size=8, seed=1).augment(x)
m = LGVae(128, 128, image_shape=[-1, 64, 64, 3], dtype="f32", device=torch.device("cuda"), seed=3); m.beta = 120.0
opt = Adam(learning_rate=1e-4); h = hashlib.sha256(); hg = hashlib.sha256()
for i in range(3):
    plan = trainer.train_step(m, img, opt); torch.cuda.synchronize()
    h.update(m.flat.detach().cpu().numpy().tobytes())
    g = m.grad_flat.detach().cpu()
    hg.update(g.numpy().tobytes())
print("fp32 weights", h.hexdigest()[:16], "grads", hg.hexdigest()[:16], "finite", bool(torch.isfinite(m.flat).all()))
PY
done
