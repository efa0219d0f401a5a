export TMPDIR=/tmp
for d in 0 1 2 3 4 7 8 15; do echo "RN_F16_DBG=$d: $(RN_F16_DBG=$d python tools/f16_head_bench.py 2>&1 | grep 'P3 alone\|200 back' | tr '\n' ' ')"; done
