for cfg in "X=1" "FI_PROPOSAL_MULTI_WG=0" "FI_CROP_NO_C1=1" "FI_NO_RING1X1=1"; do
  for rep in 1 2; do
    echo "== $cfg rep $rep"
    env $cfg python -m pytest tests/test_gpu_data_parallel.py -m gpu -q --tb=line -k "ot_l2cost-full or no_meta-small" 2>&1 | grep -E "passed|failed|^FAILED"
  done
done
