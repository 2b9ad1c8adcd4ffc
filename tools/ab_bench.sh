#!/bin/bash
# Same-box A/B of two builds of libm3vit_hip.so on the headline step (boxes differ by 2-3 %: only same-box numbers rank
# builds).   usage (on the GPU box): tools/ab_bench.sh <libA.so> <libB.so> [rounds] [extra bench.py flags]
A=$1; B=$2; R=${3:-3}; shift 3 || true
for r in $(seq 1 $R); do
  for L in "$A" "$B"; do
    ms=$(M3VIT_LIB=$L timeout -k 10 300 python bench.py --no-f32 --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $r  $L  $ms ms/step"
  done
done
