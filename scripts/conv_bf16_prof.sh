# SQ counters of the bf16 conv kernels on one layer shape (one counter group per rocprofv3 --pmc run)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
SHAPE="${1:-mask_head}"
python scripts/conv_bench.py "$SHAPE" --bf16
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
  rm -rf /tmp/p3; timeout 150 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/p3 -o c -- python scripts/conv_bench.py "$SHAPE" --bf16 > /dev/null 2>&1
  f=$(find /tmp/p3 -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'P'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Kernel_Name']
    if 'conv_bf16' in n:
        acc[(n[n.index('conv_bf16'):][:26], r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(k, len(v), sum(v)/len(v))
P
done
