"""Operator micro-benchmark on one MI355X: per-kernel time via HIP events (torch events on
the stream the kernels are launched on) for the north-star shapes.  Prints one JSON line
per operator.  Not the graded bench (that is /bench.py); used to steer kernel work and for
rocprofv3 runs (profiles/)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import clustered_dets, training_rois  # noqa: E402

DEV = "cuda:0"
HBM_PEAK = 8.0e12


def timeit(fn, iters=50, warm=10, kernel=None):
    """median/best wall time per call (HIP events around the Python call) and, when `kernel`
    names a library kernel, its mean duration from the in-library event pairs."""
    from feature_intertwiner_amd import _lib
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    if kernel:
        _lib.prof_reset()
        _lib.prof_enable(True)
    ts = []
    for _ in range(iters):
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    ts.sort()
    if kernel:
        _lib.prof_enable(False)
        n, ms = _lib.prof_get(kernel)
        KERNEL_US[0] = ms * 1e3 / max(n, 1)
    return ts[len(ts) // 2], ts[0]


KERNEL_US = [0.0]


def unique_taps_bytes(oracle_taps, C):
    """4*C*sum_r U_r with U_r = distinct (row, col) taps of RoI r."""
    yv, xv = oracle_taps["y_valid"].astype(bool), oracle_taps["x_valid"].astype(bool)
    total = 0
    for r in range(yv.shape[0]):
        rows = set(oracle_taps["y0"][r][yv[r]].tolist()) | set(oracle_taps["y1"][r][yv[r]].tolist())
        cols = set(oracle_taps["x0"][r][xv[r]].tolist()) | set(oracle_taps["x1"][r][xv[r]].tolist())
        total += len(rows) * len(cols)
    return 4 * C * total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", default="crop,cropbwd,pyramid,nhwc,roipool,nms,sinkhorn,classmean")
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    ops = args.ops.split(",")
    from feature_intertwiner_amd.roi_align.crop_and_resize import CropAndResizeFunction, pyramid_crop_and_resize
    from feature_intertwiner_amd.roi_pooling.functions.roi_pool import RoIPoolFunction
    from feature_intertwiner_amd.nms.pth_nms import nms_sorted
    from feature_intertwiner_amd.OT_module import sinkhorn_loss
    from feature_intertwiner_amd.intertwiner import class_mean, roi_level
    from oracle import oracle as O

    rs = np.random.RandomState(2000)
    B, C, S = 2, 256, 256
    image = torch.randn(B, C, S, S, device=DEV)
    rois_np = training_rois(rs, B, 256).reshape(-1, 4)
    rois = torch.from_numpy(rois_np).to(DEV)
    ind = torch.arange(B, dtype=torch.int32, device=DEV).repeat_interleave(256)
    N = rois.shape[0]

    if "crop" in ops or "cropbwd" in ops:
        for crop in (7, 14):
            fn = CropAndResizeFunction(crop, crop)
            out_bytes = 4 * N * C * crop * crop
            taps = O.crop_taps(rois_np, S, S, crop, crop)
            b_min = out_bytes + unique_taps_bytes(taps, C) + 20 * N
            b_max = out_bytes * 5
            if "crop" in ops:
                med, best = timeit(lambda: fn(image, rois, ind), args.iters, kernel="crop_fwd_%dx%d" % (crop, crop))
                k = KERNEL_US[0] * 1e-6
                print(json.dumps({"op": "crop_and_resize_fwd", "shape": [N, C, crop, crop], "map": [B, C, S, S],
                                  "us_median": med * 1e6, "us_best": best * 1e6, "kernel_us": k * 1e6,
                                  "B_min_MB": b_min / 1e6, "B_max_MB": b_max / 1e6, "GBps_Bmin": b_min / k / 1e9,
                                  "frac_hbm_Bmin": b_min / k / HBM_PEAK, "frac_hbm_Bmax": b_max / k / HBM_PEAK}))
            if "cropbwd" in ops:
                img = image.clone().requires_grad_(True)
                out = fn(img, rois, ind)
                g = torch.randn_like(out)
                med, best = timeit(lambda: torch.autograd.grad(out, img, g, retain_graph=True), args.iters,
                                   kernel="crop_bwd_%dx%d" % (crop, crop))
                print(json.dumps({"op": "crop_and_resize_bwd(+memset)", "shape": [N, C, crop, crop],
                                  "us_median": med * 1e6, "us_best": best * 1e6, "kernel_us": KERNEL_US[0]}))

    if "pyramid" in ops:
        maps = [torch.randn(4, 256, s, s, device=DEV) for s in (256, 128, 64, 32)]
        r4 = torch.from_numpy(training_rois(rs, 4, 512).reshape(-1, 4)).to(DEV)
        i4 = torch.arange(4, dtype=torch.int32, device=DEV).repeat_interleave(512)
        lv = roi_level(r4, 1024 * 1024)
        for crop in (7, 14):
            med, best = timeit(lambda: pyramid_crop_and_resize(maps, r4, i4, lv, crop, crop), args.iters,
                               kernel="crop_fwd_%dx%d" % (crop, crop))
            print(json.dumps({"op": "pyramid_crop_fwd", "shape": [r4.shape[0], 256, crop, crop],
                              "kernel_us": KERNEL_US[0],
                              "levels": np.bincount(lv.cpu().numpy(), minlength=6)[2:].tolist(),
                              "us_median": med * 1e6, "us_best": best * 1e6,
                              "GBps_write_only": 4 * r4.shape[0] * 256 * crop * crop / med / 1e9}))

    if "nhwc" in ops:
        # channels-last maps: north-star shape on ONE map (B_min from the oracle's tap table) and the
        # step's pyramid launch, forward and backward
        image_cl = image.contiguous(memory_format=torch.channels_last)
        lv1 = torch.full((N,), 2, device=DEV, dtype=torch.int32)
        for crop in (7, 14):
            out_bytes = 4 * N * C * crop * crop
            taps = O.crop_taps(rois_np, S, S, crop, crop)
            b_min = out_bytes + unique_taps_bytes(taps, C) + 20 * N
            med, best = timeit(lambda: pyramid_crop_and_resize([image_cl], rois, ind, lv1, crop, crop), args.iters,
                               kernel="crop_fwd_nhwc_%dx%d" % (crop, crop))
            k = KERNEL_US[0] * 1e-6
            print(json.dumps({"op": "crop_and_resize_fwd_nhwc", "shape": [N, C, crop, crop], "map": [B, S, S, C],
                              "us_median": med * 1e6, "kernel_us": k * 1e6, "B_min_MB": b_min / 1e6,
                              "GBps_Bmin": b_min / k / 1e9, "frac_hbm_Bmin": b_min / k / HBM_PEAK}))
            img = image_cl.clone(memory_format=torch.channels_last).requires_grad_(True)
            out = pyramid_crop_and_resize([img], rois, ind, lv1, crop, crop)
            g = torch.randn_like(out)
            med, best = timeit(lambda: torch.autograd.grad(out, img, g, retain_graph=True), args.iters,
                               kernel="crop_bwd_nhwc_%dx%d" % (crop, crop))
            print(json.dumps({"op": "crop_and_resize_bwd_nhwc(+memset)", "shape": [N, C, crop, crop],
                              "us_median": med * 1e6, "kernel_us": KERNEL_US[0]}))
        maps = [torch.randn(4, 256, s, s, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
                for s in (256, 128, 64, 32)]
        r4 = torch.from_numpy(training_rois(rs, 4, 512).reshape(-1, 4)).to(DEV)
        i4 = torch.arange(4, dtype=torch.int32, device=DEV).repeat_interleave(512)
        lv = roi_level(r4, 1024 * 1024)
        for crop in (7, 14):
            med, best = timeit(lambda: pyramid_crop_and_resize(maps, r4, i4, lv, crop, crop), args.iters,
                               kernel="crop_fwd_nhwc_%dx%d" % (crop, crop))
            kf = KERNEL_US[0]
            out = pyramid_crop_and_resize(maps, r4, i4, lv, crop, crop)
            g = torch.randn_like(out)
            medb, _ = timeit(lambda: torch.autograd.grad(out, maps, g, retain_graph=True), args.iters,
                             kernel="crop_bwd_nhwc_%dx%d" % (crop, crop))
            print(json.dumps({"op": "pyramid_crop_nhwc", "shape": [r4.shape[0], 256, crop, crop],
                              "fwd_kernel_us": kf, "fwd_us_median": med * 1e6,
                              "bwd_kernel_us": KERNEL_US[0], "bwd_us_median(+memset)": medb * 1e6}))

    if "roipool" in ops:
        pix = torch.cat([ind.float().view(-1, 1), rois[:, [1, 0, 3, 2]] * 1024.0], 1).contiguous()
        rnd = lambda v: np.floor(np.abs(v) + 0.5) * np.sign(v)          # C round(): half away from zero
        px = rois_np * 1024.0
        rw = np.maximum(rnd(px[:, 3] * 0.25) - rnd(px[:, 1] * 0.25) + 1, 1)
        rh = np.maximum(rnd(px[:, 2] * 0.25) - rnd(px[:, 0] * 0.25) + 1, 1)
        for size in (7, 14):
            fn = RoIPoolFunction(size, size, 0.25)
            med, best = timeit(lambda: fn(image, pix), args.iters, kernel="roipool_fwd")
            k = KERNEL_US[0] * 1e-6
            # out + argmax written, every element of every RoI window read once per channel
            b_alg = 4 * N * C * size * size * 2 + 4 * C * float((rw * rh).sum())
            print(json.dumps({"op": "roi_pool_fwd", "shape": [N, C, size, size], "us_median": med * 1e6,
                              "us_best": best * 1e6, "kernel_us": k * 1e6, "B_alg_MB": b_alg / 1e6,
                              "GBps": b_alg / k / 1e9, "frac_hbm": b_alg / k / HBM_PEAK,
                              "note": "window reads overlap between RoIs and the 134 MB map is Infinity-Cache "
                                      "resident, so B_alg / t may exceed the HBM figure"}))
            x = image.clone().requires_grad_(True)
            out = fn(x, pix)
            g = torch.randn_like(out)
            medb, _ = timeit(lambda: torch.autograd.grad(out, x, g, retain_graph=True), args.iters, kernel="roipool_bwd")
            print(json.dumps({"op": "roi_pool_bwd(+memset)", "shape": [N, C, size, size], "us_median": medb * 1e6,
                              "kernel_us": KERNEL_US[0]}))

    if "nms" in ops:
        for bs in (1, 4):
            dets = torch.from_numpy(np.stack([clustered_dets(rs, 6000, 1024) for _ in range(bs)])).to(DEV)
            for mk in (0, 1000):
                med, best = timeit(lambda: nms_sorted(dets, 0.7, max_keep=mk), args.iters)
                keep, num = nms_sorted(dets, 0.7, max_keep=mk)
                print(json.dumps({"op": "nms_sorted", "batch": bs, "boxes": 6000, "max_keep": mk,
                                  "kept": num.cpu().tolist(), "us_median": med * 1e6, "us_best": best * 1e6}))

    if "sinkhorn" in ops:
        for L in (5, 50):
            x = torch.relu(torch.randn(240, 256, 1, device=DEV))
            y = torch.relu(torch.randn(240, 256, 1, device=DEV))
            med, best = timeit(lambda: sinkhorn_loss(x, y, 1.0, L), args.iters)
            flop = 240 * 2 * L * 2 * 256 * 256
            print(json.dumps({"op": "sinkhorn", "problems": 240, "S": 256, "L": L, "us_median": med * 1e6,
                              "us_best": best * 1e6, "GFLOPs": flop / med / 1e9}))

    if "classmean" in ops:
        f = torch.randn(2048, 1024, device=DEV)
        gt = torch.randint(0, 81, (2048,), device=DEV, dtype=torch.int32)
        med, best = timeit(lambda: class_mean(f, gt, 81), args.iters)
        print(json.dumps({"op": "class_mean", "shape": [2048, 1024], "us_median": med * 1e6, "us_best": best * 1e6}))


if __name__ == "__main__":
    main()
