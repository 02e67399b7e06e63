#!/bin/bash
# builds libsplitvae_abl<mask>.so for each ablation mask: only wgrad_roll.hip is recompiled, the other objects are those of the shipped library
cd split_vae_amd/csrc
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -DROLL_ABL=$m -c wgrad_roll.hip -o /tmp/wgrad_roll_abl$m.o || exit 1
  objs=$(ls *.o | grep -v "_stamp\|_dbg\|_asan\|wgrad_roll" | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsplitvae_abl$m.so $objs /tmp/wgrad_roll_abl$m.o -ldl || exit 1
done
