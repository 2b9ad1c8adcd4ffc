# rocprofv3 --kernel-trace --stats of the serial (one stream, no graph) step of bench.py --config N on the GPU box:
#   tools/prof_config.sh <config> <out-dir> [extra bench flags]
# leaves <out-dir>/cfgN_kernel_stats.csv, cfgN_by_launch.txt and the bench line of the profiled run
set -e
R=$GRAFT_REPO_ROOT
C=$1; O=$R/$2; shift 2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_cfg$C
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg$C -o run -- python $R/bench.py --config $C --steps 4 --warmup 1 --no-cpu-baseline --serial-tasks --no-graph "$@" > $O/cfg${C}_under_rocprof.json 2> $O/cfg${C}_prof.err
cp $(find /tmp/prof_cfg$C -name "*kernel_stats.csv" | head -1) $O/cfg${C}_kernel_stats.csv
python $R/tools/prof_by_launch.py $(find /tmp/prof_cfg$C -name "*kernel_trace.csv" | head -1) --steps 9 > $O/cfg${C}_by_launch.txt
rm -rf /tmp/prof_cfg$C
