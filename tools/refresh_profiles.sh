# Re-creates the measurement artifacts behind profiles/ on the GPU box (copy the summaries into profiles/ afterwards):
#   tools/refresh_profiles.sh [<commit for the traffic file's "head" field>]
set -e
R=$GRAFT_REPO_ROOT
HEAD=${1:-unknown}
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err
B="--no-f32 --no-cpu-baseline"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -o run -- python $R/bench.py --steps 8 --warmup 2 $B > $O/prof2.json 2> $O/prof2.err
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o run -- python $R/bench.py --steps 8 --warmup 2 $B --serial-tasks --no-graph > $O/prof1.json 2> $O/prof1.err
mkdir -p $O/fetch $O/write
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o run -- python $R/bench.py --steps 2 --warmup 1 $B --serial-tasks --no-graph > $O/fetch.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o run -- python $R/bench.py --steps 2 --warmup 1 $B --serial-tasks --no-graph > $O/write.log 2>&1
python $R/tools/pmc_traffic.py $(find $O/fetch -name "*.db" | head -1) $(find $O/write -name "*.db" | head -1) --dtype f16 --head $HEAD \
  --merge gemm_nt_all=gemm_nt_dma,gemm_nt > $O/pmc_traffic.json
cp $(find $O/prof2 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_two_streams.csv
cp $(find $O/prof1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial.csv
# the databases are large: keep only the summaries in what gpurun copies back
rm -rf $O/fetch $O/write $O/prof1 $O/prof2
ls -la $O
echo refresh done
