# PMC passes over the conv micro-benchmark (one counter group per rocprofv3 run, each under timeout)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
SHAPE="${1:-FPN P2}"
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  rm -rf /tmp/p3; timeout 150 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/p3 -o c -- python scripts/conv_bench.py "$SHAPE" > /dev/null 2>&1
  f=$(find /tmp/p3 -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'P'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Kernel_Name']
    if 'conv' in n:
        acc[(n[28:62], r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(k, len(v), sum(v)/len(v))
P
done
