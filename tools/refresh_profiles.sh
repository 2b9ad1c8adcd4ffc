set -e
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/final; mkdir -p $R/gpurun_out/final
cd $R
timeout -k 10 600 python bench.py > gpurun_out/final/bench_f16.json 2> gpurun_out/final/bench_f16.err
timeout -k 10 600 python bench.py --dtype f32 --no-cpu-baseline > gpurun_out/final/bench_f32.json 2> gpurun_out/final/bench_f32.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof2 -o run -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/gpurun_out/final/prof2.json 2> $R/gpurun_out/final/prof2.err
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof1 -o run -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --serial-tasks --no-graph > $R/gpurun_out/final/prof1.json 2> $R/gpurun_out/final/prof1.err
mkdir -p $R/gpurun_out/final/fetch $R/gpurun_out/final/write
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/final/fetch -o run -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial-tasks --no-graph > $R/gpurun_out/final/fetch.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/final/write -o run -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --serial-tasks --no-graph > $R/gpurun_out/final/write.log 2>&1
find $R/gpurun_out/final -name "*.db" -o -name "*stats*.csv" | head -20
echo refresh done
