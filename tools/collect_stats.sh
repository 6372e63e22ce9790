#!/bin/bash
# rocprofv3 kernel stats of the bench command (graph replay + eager event pass) -> gpurun_out/$1/
set -e
V=${1:-stats}
shift || true
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
O=$ROOT/gpurun_out/$V
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --repeat 2 "$@" > $O/stats.log 2>&1
find $O -name "*kernel_stats.csv" | head -3
