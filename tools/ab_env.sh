#!/bin/bash
# same-box A/B of an environment switch on the headline step: tools/ab_env.sh VAR rounds val1 val2 ... [-- bench flags]
V=$1; R=$2; shift 2
VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done; [ "$1" == "--" ] && shift
for r in $(seq 1 $R); do
  for X in "${VALS[@]}"; do
    ms=$(env $V=$X timeout -k 10 300 python bench.py --no-f32 --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $r  $V=$X  $ms ms/step"
  done
done
