#!/bin/bash
# launch-shape sweep of the net kernels and the forward-branch knob (one GPU call)
cd "${GRAFT_REPO_ROOT:-.}"
B="python bench.py --no-cpu-baseline --no-also"
O=gpurun_out/sweep1; mkdir -p $O
$B > $O/base.json 2>/dev/null
PF_FWD_SERIAL=1 $B > $O/fwd_serial.json 2>/dev/null
for fb in 1024 1536 3072 4096; do PF_FWD_BLOCKS=$fb $B > $O/fwdblocks_$fb.json 2>/dev/null; done
for pb in 512 768; do PINNFEM_PART_BLOCKS=$pb $B > $O/partblocks_$pb.json 2>/dev/null; done
python tools/show_bench.py $O/*.json
