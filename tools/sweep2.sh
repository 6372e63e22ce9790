#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
B="python bench.py --no-cpu-baseline --no-also"
O=gpurun_out/sweep2; mkdir -p $O
$B > $O/base.json 2>/dev/null
PF_FWD_T=2 $B > $O/fwdT2.json 2>$O/fwdT2.err
PF_FWD_T=2 PF_FWD_BLOCKS=1024 $B > $O/fwdT2_b1024.json 2>/dev/null
PF_FWD_T=2 PF_FWD_BLOCKS=512 $B > $O/fwdT2_b512.json 2>/dev/null
PF_FWD_BLOCKS=512 $B > $O/fwdT1_b512.json 2>/dev/null
PF_FWD_BLOCKS=256 $B > $O/fwdT1_b256.json 2>/dev/null
python tools/show_bench.py $O/*.json
PF_FWD_T=2 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "single_step or first_adam" 2>&1 | tail -2
