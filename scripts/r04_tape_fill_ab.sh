# the tape's step-start zero fills: one 16-B-store launch vs two hipMemsetAsync calls
# (experiment of round 4, NOT kept: the one-launch fill and its SV_TAPE_MEMSET knob were removed again -- LAB_NOTES 4k; kept for the record of how profiles/r04_h_tape_fill_ab.txt was made)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_spair_model.py -q -x -m gpu 2>&1 | grep -E "passed|failed" | tail -2
for r in 1 2 3; do for dt in f32 bf16; do for v in BASE=1 SV_TAPE_MEMSET=1; do echo -n "$dt $v: "; env $v python scripts/bench_spair_native.py 32 $dt 2>/dev/null | tail -1; done; done; done
