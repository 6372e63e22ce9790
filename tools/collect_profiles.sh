#!/bin/bash
# One GPU call: the bench line, the rocprofv3 kernel stats of the same command, the two HBM-traffic PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate, --kernel-trace only) and the parity summary -> gpurun_out/prof_$1/ ; copy what is
# judged to profiles/.
set -e
V=${1:-v5}
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
O=$ROOT/gpurun_out/prof_$V
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo bench done
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-also --repeat 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $ROOT/bench.py $ARGS > $O/stats.log 2>&1
echo stats done
PARGS="--steps 10 --warmup 10 --no-cpu-baseline --no-also --repeat 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 $ROOT/bench.py $PARGS > $O/fetch.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 $ROOT/bench.py $PARGS > $O/write.log 2>&1
echo write done
python3 $ROOT/tools/pmc_traffic.py $(find $O/fetch -name "*counter_collection.csv") $(find $O/write -name "*counter_collection.csv") $O/traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py $PARGS; FETCH_SIZE doubled (gfx950)" > $O/traffic_per_launch.txt
python3 $ROOT/tools/parity_summary.py > $O/parity_summary.json 2> $O/parity.err
echo parity done
