#!/bin/bash
export SV_BENCH_OPS=wgrad
for m in "$@"; do echo -n "abl=$m  "; SV_LIB_NAME=libsplitvae_abl$m.so timeout 120 python scripts/bench_layers.py 1024 d4 2>&1 | grep d4; done
