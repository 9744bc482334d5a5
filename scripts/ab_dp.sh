#!/bin/bash
# the data-parallel engine's own cost on one GPU (FI_DP_FORCE=1: every collective runs in a 1-rank RCCL group), A/B of
# its switches inside ONE box call, interleaved `rounds` times:  bash scripts/ab_dp.sh 2 "X=1" "FI_META_SIDE_DP=0" ...
R="$1"; shift
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for i in $(seq $R); do
  echo -n "plain : "; python bench.py --no-pmc --no-cpu-baseline --no-dense-reference --steps 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  for v in "$@"; do
    echo -n "DP $v : "; env FI_DP_FORCE=1 $v python bench.py --no-pmc --no-cpu-baseline --no-dense-reference --steps 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('data_parallel',{}).get('buckets'))"
  done
done
