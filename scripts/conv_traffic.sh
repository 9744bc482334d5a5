# FETCH_SIZE / WRITE_SIZE (KB as reported; FETCH x2 on gfx950 for 16-byte reads) of the conv kernels on one layer shape
#   bash scripts/conv_traffic.sh "FPN P2" [ENV=VALUE ...]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
SHAPE="${1:-FPN P2}"; shift
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p4; env "$@" timeout 150 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/p4 -o c -- python scripts/conv_bench.py "$SHAPE" > /dev/null 2>&1
  f=$(find /tmp/p4 -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'P'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Kernel_Name']
    if 'conv' in n:
        acc[(n.split('(')[0][-48:], r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(k, len(v), "mean KB %.0f" % (sum(v)/len(v)))
P
done
