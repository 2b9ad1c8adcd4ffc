# ONE two-rank rehearsal per setting (two PROCESSES on one MI355X over gloo, data-parallel leg with one linear
# hipGraph per task pass - the form that stalled for seconds per step in round 4) with the number of HIP streams per process
# varied: (a) two task streams per process as benchmarked at N = 1, (b) the task passes folded onto ONE stream
# (--serial-tasks), (c) two task streams with the runtime's hardware-queue limit raised (GPU_MAX_HW_QUEUES=8).
#   tools/dp_two_rank_streams.sh <out-file>
R=$GRAFT_REPO_ROOT
OUT=$R/$1
export M3_BENCH_BACKEND=gloo M3_BENCH_ONE_DEVICE=1 M3_LINEAR_GRAPHS=1 M3_BENCH_NO_AGREE=0
run() {
  tag=$1; envs=$2; flags=$3
  echo "=== $tag: env $envs flags $flags" >> $OUT
  ( cd $R && timeout -k 10 240 env $envs python bench.py --gpus 2 --dp-only --steps 20 --warmup 3 --no-cpu-baseline $flags > /tmp/two_rank.json 2> /tmp/two_rank.err )
  echo "rc=$?" >> $OUT
  python - >> $OUT <<PY
import json
try:
    d = json.load(open("/tmp/two_rank.json"))
    print({k: d.get(k) for k in ("value", "ms_per_step")}, d.get("dp"), d["config"].get("launch"), d["config"].get("task_streams"))
except Exception as e:
    print("no line:", e)
PY
  grep -h "timed region\|host launch\|FAILED\|Error" /tmp/two_rank.err | tail -6 >> $OUT
}
: > $OUT
run "(a) two task streams per process, linear graphs, 4 hardware queues (HIP default)" "GPU_MAX_HW_QUEUES=4" ""
run "(b) task passes folded onto ONE stream per process, 4 queues" "GPU_MAX_HW_QUEUES=4" "--serial-tasks"
run "(c) two task streams, GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=8" ""
run "(d) two task streams, one graph per part (M3_LINEAR_GRAPHS=0), 4 queues" "M3_LINEAR_GRAPHS=0 GPU_MAX_HW_QUEUES=4" ""
run "(e) two task streams, one graph per part (M3_LINEAR_GRAPHS=0), 8 queues" "M3_LINEAR_GRAPHS=0 GPU_MAX_HW_QUEUES=8" ""
