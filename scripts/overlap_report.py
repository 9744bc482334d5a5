"""Does the gradient all-reduce overlap backward?  Timeline report from a rocprofv3 trace of a
multi-rank bench run (kernel trace + memory-copy trace, one set of CSV files per process).

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- \
        python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 ...
    python scripts/overlap_report.py <dir>

Per rank and per train step it prints the backward window (first fused-backward kernel .. last
weight-gradient kernel of the step), and for every gradient bucket's transfer (on a one-GPU box the
ranks talk over gloo: device->host copy of the flat bucket, host reduction, host->device copy;
over RCCL these are the ring kernels) whether it started before backward ended.  The statement
proved: buckets are issued from autograd hooks DURING backward, on a side stream."""
import csv
import glob
import os
import sys


def read(path):
    with open(path) as f:
        return list(csv.DictReader(f))


def col(rows, *names):
    for n in names:
        if rows and n in rows[0]:
            return n
    raise KeyError("none of %r in %r" % (names, list(rows[0].keys()) if rows else []))


def main():
    root = sys.argv[1]
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0      # the trace carries no byte counts
    ktraces = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))
    print("# %d kernel traces under %s" % (len(ktraces), root))
    for kt in ktraces:
        ks = read(kt)
        if len(ks) < 1000:
            continue
        mc_path = kt.replace("kernel_trace.csv", "memory_copy_trace.csv")
        ms = read(mc_path) if os.path.exists(mc_path) else []
        kn, k0, k1 = col(ks, "Kernel_Name"), col(ks, "Start_Timestamp"), col(ks, "End_Timestamp")
        bwd = [(int(r[k0]), int(r[k1]), r[kn]) for r in ks
               if ("bn_act_bwd" in r[kn] or "conv_wgrad" in r[kn] or "crop_bwd" in r[kn])]
        # torch._utils._flatten_dense_tensors of a bucket = one CatArrayBatchedCopy launch: the moment
        # GradientBuckets hands a bucket to the communication stream
        flat = [(int(r[k0]), int(r[k1])) for r in ks if "CatArrayBatchedCopy" in r[kn]]
        fwd_marks = sorted(int(r[k0]) for r in ks if "nms_mask_kernel" in r[kn])     # one NMS per step
        if not bwd or not fwd_marks:
            continue
        print("\n## process trace %s: %d kernels, %d memory copies, %d steps" % (
            os.path.basename(kt), len(ks), len(ms), len(fwd_marks)))
        copies = []
        if ms:
            m0, m1, md = col(ms, "Start_Timestamp"), col(ms, "End_Timestamp"), col(ms, "Direction", "Kind")
            copies = [(int(r[m0]), int(r[m1]), r[md].replace("MEMORY_COPY_", "")) for r in ms]
        bounds = fwd_marks + [1 << 62]
        for s in range(len(fwd_marks)):
            lo, hi = bounds[s], bounds[s + 1]
            b = [x for x in bwd if lo <= x[0] < hi]
            if not b:
                continue
            b_start, b_end = min(x[0] for x in b), max(x[1] for x in b)
            fl = [f for f in flat if b_start <= f[0] < hi]
            fl_in = [f for f in fl if f[0] < b_end]
            cs = [c for c in copies if b_start <= c[0] < hi and (c[1] - c[0]) >= min_us * 1e3]
            cs_in = [c for c in cs if c[0] < b_end]
            print("step %d: backward window %.1f ms | bucket flatten launches: %d, of which inside the backward window: %d "
                  "| copies >= %.0f us after backward started: %d, inside the window: %d" % (
                      s, (b_end - b_start) / 1e6, len(fl), len(fl_in), min_us, len(cs), len(cs_in)))
            if s == len(fwd_marks) - 1:
                for f in fl:
                    print("    bucket flatten at %+8.2f ms relative to the END of backward" % ((f[0] - b_end) / 1e6))
                for c in cs:
                    print("    %-16s %6.0f us  start %+8.2f ms relative to the END of backward" % (
                        c[2], (c[1] - c[0]) / 1e3, (c[0] - b_end) / 1e6))


if __name__ == "__main__":
    main()
