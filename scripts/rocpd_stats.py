"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / mean / min /
max duration.  Usage: python scripts/rocpd_stats.py results.db [substring ...]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    filt = sys.argv[2:]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = c.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(n, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values()) or 1
    print("%-72s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "mean_us", "min_us", "max_us", "pct"))
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if filt and not any(f in n for f in filt):
            continue
        print("%-72s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (n[:72], a[0], a[1] / 1e3, a[1] / a[0] / 1e3,
                                                             a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / tot))


if __name__ == "__main__":
    main()
