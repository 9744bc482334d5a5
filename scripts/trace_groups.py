"""Group a rocprofv3 kernel trace by (kernel, grid): count, mean duration, total -- the per-layer view of
one kernel (grid size identifies the layer shape).  usage: trace_groups.py <kernel_trace.csv> [substr] [steps]"""
import csv, re, sys, collections
path = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
g = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"]
    if sub and sub not in n:
        continue
    n = re.sub(r"\(anonymous namespace\)::|void ", "", n).split("(")[0][:60]
    key = (n, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    g[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    tot += sum(v)
    print("%-60s grid %6d x %4d x %3d  n/step %6.1f  mean %9.1f us  min %9.1f  ms/step %8.3f" % (
        k[0], k[1], k[2], k[3], len(v) / steps, sum(v) / len(v), min(v), sum(v) / 1e3 / steps))
print("total ms/step %.3f" % (tot / 1e3 / steps))
