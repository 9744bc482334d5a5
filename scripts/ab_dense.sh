#!/bin/bash
# The backward pass in its dense (round-2) form against the default, interleaved inside ONE box call: ms/step of the fp32
# headline workload and of cfg5 on the bf16 kernels.
for cfgargs in "" "--config cfg5"; do
  for i in 1 2; do
    for v in "--dense-backward" ""; do
      echo -n "bench.py $cfgargs ${v:-(default)} : "; python bench.py --no-pmc --no-cpu-baseline --no-dense-reference --steps 12 $cfgargs $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
    done
  done
done
