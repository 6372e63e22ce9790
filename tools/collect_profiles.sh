#!/bin/bash
# One GPU call: the bench line, the rocprofv3 kernel stats of the same command and the two PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate, --kernel-trace only) -> gpurun_out/prof_$1/ ; copy what is judged to profiles/.
set -e
V=${1:-v5}
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
O=$ROOT/gpurun_out/prof_$V
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo bench done
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-also"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $ROOT/bench.py $ARGS > $O/stats.log 2>&1
echo stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 $ROOT/bench.py --steps 10 --warmup 10 --no-cpu-baseline --no-also > $O/fetch.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 $ROOT/bench.py --steps 10 --warmup 10 --no-cpu-baseline --no-also > $O/write.log 2>&1
echo write done
find $O -name "*.csv" | head -20
