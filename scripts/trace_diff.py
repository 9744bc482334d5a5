"""Two rocprofv3 kernel traces of the train step side by side: per kernel name, launches and milliseconds per step in
each, sorted by the difference -- what one variant of the step runs that the other does not.
    python scripts/trace_diff.py A_kernel_trace.csv B_kernel_trace.csv [steps=4]"""
import csv
import sys
from collections import defaultdict


def load(path, steps):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [s for s, e, n in rows if "sgd_update_kernel" in n]
    lo, hi = marks[-steps - 1], marks[-1]
    tot, cnt = defaultdict(float), defaultdict(int)
    for s, e, n in rows:
        if s >= lo and e <= hi:
            k = n.replace("void ", "").replace("(anonymous namespace)::", "")
            k = k.split("(")[0][:70]
            tot[k] += (e - s) / 1e6 / steps
            cnt[k] += 1
    return tot, cnt, (hi - lo) / 1e6 / steps


steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ta, ca, wa = load(sys.argv[1], steps)
tb, cb, wb = load(sys.argv[2], steps)
print("wall ms/step: A %.2f  B %.2f ; kernel time summed over streams: A %.2f  B %.2f" % (wa, wb, sum(ta.values()), sum(tb.values())))
keys = sorted(set(ta) | set(tb), key=lambda k: -abs(tb.get(k, 0) - ta.get(k, 0)))
for k in keys[:40]:
    print("%-72s A %7.3f ms %5.1f x   B %7.3f ms %5.1f x   diff %+7.3f" % (k, ta.get(k, 0), ca.get(k, 0) / steps, tb.get(k, 0),
                                                                       cb.get(k, 0) / steps, tb.get(k, 0) - ta.get(k, 0)))
