#!/bin/bash
# Same-box A/B of two source TREES on the headline step (python + kernels both differ; boxes differ by 2-3 %: only
# same-box numbers rank builds).  usage (on the GPU box): tools/ab_trees.sh <treeA> <treeB> [rounds] [extra bench.py flags]
# A tree is a checkout with its own built m3vit_amd/libm3vit_hip.so, e.g. made here by
#   git archive <commit> | tar -x -C build/base && make -C build/base/m3vit_amd/csrc   (build/ travels with gpurun)
A=$1; B=$2; R=${3:-3}; shift 3 || true
for r in $(seq 1 $R); do
  for T in "$A" "$B"; do
    ms=$(cd $T && timeout -k 10 300 python bench.py --no-f32 --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $r  $T  $ms ms/step"
  done
done
