#!/bin/bash
# run-to-run repeatability of the default backward form under the A/B switches that change the stream timing
# (scripts/backward_forms_probe.py --self): one number per pass = the largest relative deviation over the parameters
out=gpurun_out/${1:-r04}_bfp_sweep.txt
: > $out
run() { echo "=== $*" >> $out; env "$@" python scripts/backward_forms_probe.py --steps 4 --repeat ${REP:-24} --self 2>&1 | grep "^pass" | awk '{print $6}' | tr '\n' ' ' >> $out; echo >> $out; }
run X=1
run FI_PROPOSAL_SIDE=0
run FI_DEAD_SIDE=0
run FI_STATIC_DEV=0
run FI_STATIC_DEV=0 FI_DEAD_SIDE=0
