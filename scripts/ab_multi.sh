#!/bin/bash
# A/B/C... of env switches inside ONE box call: bench each variant, interleaved, `rounds` times.
#   bash scripts/ab_multi.sh 2 "FI_X=0" "FI_X=1 FI_Y=0" ...
R="$1"; shift
for i in $(seq $R); do
  for v in "$@"; do
    echo -n "$v : "; env $v python bench.py --no-pmc --no-cpu-baseline --no-dense-reference --steps 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  done
done
