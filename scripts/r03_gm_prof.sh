#!/bin/bash
# rocprofv3 kernel-trace summary + one step's timeline of the SPLIT-GMVAE train step (config 3: SVHN-32, B = 64, bf16) -> gpurun_out/<tag>_gm_*
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_gm
cat > /tmp/gm_drv.py <<PY
import sys, time
sys.path.insert(0, "$ROOT")
import torch
from split_vae_amd import data
from split_vae_amd.augmentation import Augmentator
from split_vae_amd.gm import LGGMVae, train_step_lg_gm_vae
from split_vae_amd.optimizer import Adam
x = data.synthetic_images(64, 32, 32, seed=0, device="cuda")
aug = Augmentator("scramble", size=4, seed=1)
m = LGGMVae(128, 128, [-1, 32, 32, 3], 30, 0.4, dtype=__import__("os").environ.get("GM_DTYPE", "bf16"), device="cuda", seed=3)
m.beta, m.alpha = 40.0, 40.0
opt = Adam(learning_rate=1e-4)
for _ in range(10): train_step_lg_gm_vae(m, aug.augment(x, plan=m.plan(64)), opt)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): train_step_lg_gm_vae(m, aug.augment(x, plan=m.plan(64)), opt)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue %.3f ms/step, wall %.3f ms/step" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
PY
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_gm -o gm --output-format csv -- python3 /tmp/gm_drv.py > /tmp/prof_gm.log 2>&1
tail -2 /tmp/prof_gm.log | tee $ROOT/gpurun_out/${TAG}_gm_host.txt
f=$(find /tmp/prof_gm -name '*kernel_stats.csv' | head -1)
head -45 "$f" | cut -c1-200 > $ROOT/gpurun_out/${TAG}_gm_kernel_stats.txt
python3 - "$f" <<'PY' | tee -a $ROOT/gpurun_out/${TAG}_gm_kernel_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
calls = sum(int(r["Calls"]) for r in rows)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("# steps 30 (10 warm-up + 20 timed): %.1f launches and %.3f ms of kernel time per step" % (calls / 30.0, tot / 30.0 / 1e6))
PY
t=$(find /tmp/prof_gm -name '*kernel_trace.csv' | head -1)
python3 $ROOT/scripts/timeline.py "$t" > $ROOT/gpurun_out/${TAG}_gm_timeline.txt
