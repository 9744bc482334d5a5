#!/bin/bash
# A/B of a Python-level switch inside ONE box call: bench twice each way, interleaved.
#   bash scripts/ab_env.sh "FI_WGRAD_SIDE=0" "FI_WGRAD_SIDE=1" [bench args...]
A="$1"; B="$2"; shift 2
for i in 1 2; do
  for v in "$A" "$B"; do
    echo -n "$v : "; env $v python bench.py --no-pmc --no-cpu-baseline --no-dense-reference --steps 12 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  done
done
