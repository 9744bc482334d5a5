cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o c -- python scripts/op_bench.py --ops crop,pyramid > /dev/null 2>&1
grep -E "crop_fwd" /tmp/p1/c_kernel_stats.csv | awk -F'","' '{print substr($1,1,70), "calls",$2, "avg",$4, "min",$6, "max",$7}'
for ctr in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE TA_TA_BUSY_sum" "TCP_TA_TCP_STATE_READ_sum TCP_TAGCONFLICT_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  rm -rf /tmp/p2; timeout 150 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/p2 -o c -- python scripts/op_bench.py --ops crop --iters 5 > /dev/null 2>&1
  f=$(find /tmp/p2 -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'P'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'crop_fwd' in r['Kernel_Name']:
        acc[(r['Kernel_Name'][28:60], r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(k, len(v), sum(v)/len(v))
P
done
