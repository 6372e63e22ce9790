#!/bin/bash
# SQ-level PMC passes over tools/slots.py (separate passes, --kernel-trace only).  usage: pmc_slots.sh <tag> [slots.py args]
set -e
TAG=${1:-x}; shift || true
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- python3 $ROOT/tools/slots.py "$@" > $OUT/p1.log 2>&1
echo pass1 done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --output-format csv -d $OUT/p2 -- python3 $ROOT/tools/slots.py "$@" > $OUT/p2.log 2>&1
echo pass2 done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/p3 -- python3 $ROOT/tools/slots.py "$@" > $OUT/p3.log 2>&1 || echo pass3 failed
echo pass3 done
python3 $ROOT/tools/pmc_summary.py $(find $OUT -name "*counter_collection.csv" | sort) > $OUT/summary.txt
cat $OUT/summary.txt | cut -c1-600
