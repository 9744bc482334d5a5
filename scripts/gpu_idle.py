"""GPU idle time of the train step from a rocprofv3 kernel trace: the union of all kernels' [start, end) intervals (every
stream) against the wall time of the traced steps, and the histogram of the gaps.
    rocprofv3 --kernel-trace --output-format csv -d /tmp/px -o kt -- python bench.py --steps 6 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-pmc --no-dense-reference
    python scripts/gpu_idle.py /tmp/px/.../kt_kernel_trace.csv"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the timed steps: the last 6 occurrences of the optimiser's update kernel delimit steps
marks = [s for s, e, n in rows if "sgd_update_kernel" in n]
lo, hi = marks[-5], marks[-1]          # 4 whole steps
iv = [(s, e) for s, e, n in rows if s >= lo and e <= hi]
busy, gaps, cur_e = 0, [], None
for s, e in iv:
    if cur_e is None:
        cur_s, cur_e = s, e
    elif s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
busy += cur_e - cur_s
wall = hi - lo
print("4 steps: wall %.2f ms/step, GPU busy (union over streams) %.2f ms/step, idle %.2f ms/step in %d gaps/step" % (
    wall / 4e6, busy / 4e6, (wall - busy) / 4e6, len(gaps) // 4))
for lim in (2, 5, 10, 20, 50, 100, 1000, 1e9):
    sel = [g for g in gaps if g / 1e3 <= lim]
    print("  gaps <= %6.0f us: %5d per step, %.3f ms per step" % (lim, len(sel) // 4, sum(sel) / 4e6))
print("largest gaps (us):", sorted([round(g / 1e3, 1) for g in gaps])[-12:])
