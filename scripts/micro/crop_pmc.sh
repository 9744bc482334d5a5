#!/bin/bash
# PMC passes over the RoIAlign micro-benchmark (one counter group per pass; --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
BIN=$GRAFT_REPO_ROOT/scripts/micro/bin/crop_var
OUT=$GRAFT_REPO_ROOT/gpurun_out/crop_pmc
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -o "\b\(TCP\|TA\|TCC\|TD\|SQ\|GRBM\)_[A-Z0-9_a-z]*" | sort -u > $OUT/counters.txt
SEL="${1:-library}"
i=0
for grp in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE GRBM_COUNT" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -o p -- $BIN "$SEL" > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY' >> $OUT/summary.txt
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k,c),v in sorted(acc.items()):
    # a dispatch contributes several rows (one per dimension instance) -- sum per dispatch is not available here; print mean*rows/dispatches
    print("%-62s %-40s n=%4d mean=%.1f sum=%.1f" % (k,c,len(v),sum(v)/len(v),sum(v)))
PY
  else echo "group $i: no counter file ($grp)" >> $OUT/summary.txt; tail -3 $OUT/p$i.log >> $OUT/summary.txt; fi
  rm -rf $OUT/p$i
done
