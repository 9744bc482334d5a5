# HBM-side traffic of the dominant kernels: FETCH_SIZE and WRITE_SIZE (KB) in SEPARATE rocprofv3
# --pmc passes (the guide's recipe; both in one pass hung on this pool), each under a short timeout.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() {  # $1 = label, $2 = counter, rest = command
  label="$1"; ctr="$2"; shift; shift
  rm -rf /tmp/p4; timeout 90 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/p4 -o c -- "$@" > /dev/null 2>&1
  rc=$?
  f=$(find /tmp/p4 -name '*counter_collection.csv' 2>/dev/null | head -1)
  if [ -z "$f" ]; then echo "$label | $ctr | no output (rc=$rc)"; return; fi
  python - "$f" "$label" <<'P'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Kernel_Name']
    if any(k in n for k in ('conv_fwd_kernel<128, 3, 3','conv_wgrad_vec_kernel<128, 3, 3','crop_fwd_cl_kernel','crop_bwd_cl_kernel','crop_fwd_kernel<7')):
        acc[(n[n.index('::')+2:][:48], r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()):
    print(sys.argv[2], '|', k[0], '|', k[1], '| launches', len(v), '| mean KB', round(sum(v)/len(v),1), '| max KB', round(max(v),1))
P
}
for c in FETCH_SIZE WRITE_SIZE; do
  run "op_bench nhwc (512 RoIs on one map, then 2048 RoIs pyramid)" $c python scripts/op_bench.py --ops nhwc --iters 3
  run "conv_bench FPN P2 smooth 3x3 (4x256x256x256 -> 256)" $c python scripts/conv_bench.py "FPN P2"
done
