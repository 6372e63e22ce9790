#!/bin/bash
# matrix-pipe / vector-ALU co-execution counters of the MLP kernels (separate PMC pass, --kernel-trace only)
set -e
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
O=$ROOT/gpurun_out/${1:-pmc_coexec}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 10 --no-cpu-baseline --no-also --repeat 0"
rocprofv3 -L > $O/counters_list.txt 2>&1 || true
grep -o "SQ_[A-Z_]*MFMA[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_VALU[A-Z_]*" $O/counters_list.txt | sort -u > $O/mfma_counters.txt || true
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- python3 $ROOT/bench.py $ARGS > $O/p1.log 2>&1 || echo pass1 failed
echo pass1 done
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/p2 -- python3 $ROOT/bench.py $ARGS > $O/p2.log 2>&1 || echo pass2 failed
echo pass2 done
python3 $ROOT/tools/pmc_summary.py $(find $O -name "*counter_collection.csv") > $O/summary.txt 2>&1 || true
