# Re-creates the measurement artifacts behind profiles/ on the GPU box (copy the summaries into profiles/ afterwards):
#   tools/refresh_profiles.sh [<commit for the traffic files' "head" field>]
# f16 (headline) AND f32 (train_fastmoe.py's arithmetic): bench lines, rocprofv3 --kernel-trace --stats summaries of the
# two-stream default and of the serial (--serial-tasks --no-graph) step, and the FETCH_SIZE / WRITE_SIZE PMC passes
# (separate runs, --kernel-trace only, program directly after `--`).
set -e
R=$GRAFT_REPO_ROOT
HEAD=${1:-unknown}
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err
B="--no-f32 --no-cpu-baseline --no-vitb --no-skew --no-module-path"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -o run -- python $R/bench.py --steps 8 --warmup 2 $B > $O/prof2.json 2> $O/prof2.err
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o run -- python $R/bench.py --steps 8 --warmup 2 $B --serial-tasks --no-graph > $O/prof1.json 2> $O/prof1.err
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1f32 -o run -- python $R/bench.py --dtype f32 --steps 4 --warmup 1 --no-cpu-baseline --no-vitb --no-skew --no-module-path --serial-tasks --no-graph > $O/prof1_f32.json 2> $O/prof1_f32.err
cp $(find $O/prof2 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_two_streams.csv
cp $(find $O/prof1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial.csv
cp $(find $O/prof1f32 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_f32.csv
python $R/tools/prof_by_launch.py $(find $O/prof1 -name "*kernel_trace.csv" | head -1) --steps 14 > $O/by_launch_shape_serial.txt
python $R/tools/prof_by_launch.py $(find $O/prof1f32 -name "*kernel_trace.csv" | head -1) --steps 9 > $O/by_launch_shape_serial_f32.txt
rm -rf $O/prof1 $O/prof2 $O/prof1f32
# per-SHAPE tables of the two GEMM entry points (HIP events around every launch of a serial eager step)
cd $R
for spec in "f16 1" "f32 1" "f16 3"; do
  set -- $spec
  timeout -k 10 300 python tools/shape_times.py --dtype $1 --config $2 > $O/shape_times_$1_cfg$2.txt 2>&1
done
cd /tmp
# the ViT-Base configurations (single-GPU forms): serial traces
$R/tools/prof_config.sh 3 gpurun_out/final && $R/tools/prof_config.sh 4 gpurun_out/final
cd /tmp
for DT in f16 f32; do
  mkdir -p $O/fetch_$DT $O/write_$DT
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch_$DT -o run -- python $R/bench.py --dtype $DT --steps 2 --warmup 1 $B --serial-tasks --no-graph > $O/fetch_$DT.log 2>&1
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write_$DT -o run -- python $R/bench.py --dtype $DT --steps 2 --warmup 1 $B --serial-tasks --no-graph > $O/write_$DT.log 2>&1
  python $R/tools/pmc_traffic.py $(find $O/fetch_$DT -name "*.db" | head -1) $(find $O/write_$DT -name "*.db" | head -1) --dtype $DT --head $HEAD \
    --merge gemm_nt_all=gemm_nt_dma,gemm_nt > $O/pmc_traffic_$DT.json
  # the databases are large: keep only the summaries in what gpurun copies back
  rm -rf $O/fetch_$DT $O/write_$DT
done
# the ViT-Base configurations' counters (bench.py replays them as configs3 / configs4 .roofline.traffic:
# copy to profiles/rNN_cfg3_hbm_bytes_per_launch.json / rNN_cfg4_hbm_bytes_per_launch.json)
cd $R
tools/pmc_config.sh 3 gpurun_out/final $HEAD
tools/pmc_config.sh 4 gpurun_out/final $HEAD
ls -la $O
echo refresh done
