# FETCH_SIZE / WRITE_SIZE passes (separate runs, --kernel-trace only) over the serial step of bench.py --config N:
#   tools/pmc_config.sh <config> <out-dir> [head]
set -e
R=$GRAFT_REPO_ROOT
C=$1; O=$R/$2; HEAD=${3:-unknown}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--config $C --no-f32 --no-cpu-baseline --no-vitb --no-skew --no-module-path --steps 2 --warmup 1 --serial-tasks --no-graph"
rm -rf /tmp/pf_$C /tmp/pw_$C
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf_$C -o run -- python $R/bench.py $B > $O/fetch_cfg$C.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw_$C -o run -- python $R/bench.py $B > $O/write_cfg$C.log 2>&1
python $R/tools/pmc_traffic.py $(find /tmp/pf_$C -name "*.db" | head -1) $(find /tmp/pw_$C -name "*.db" | head -1) --dtype f16 --head $HEAD > $O/pmc_traffic_cfg$C.json
rm -rf /tmp/pf_$C /tmp/pw_$C
