"""Headline benchmark: images/sec of the full Feature-Intertwiner train step
(BASELINE.json: ResNet-101-FPN, 1024x1024, 512 RoIs/image, OT intertwiner on, Sinkhorn
L=50, 80 classes) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; weak scaling (4 images per GPU at every N); rank 0 prints ONE JSON
line.  A step = forward + intertwiner (OT) loss + backward + gradient all-reduce + clip
+ SGD update on one synthetic COCO-shaped batch already resident in HBM (synthetic.py).
All arithmetic is fp32, as in the reference.

Extra objects on the JSON line
  roofline      for crop_fwd_kernel<7, 7> (RoIAlign 7x7, the north-star kernel): algorithmic
                bytes (SURVEY 8d's B_min = output + unique taps per RoI + box records) of its
                launch in the step / its mean duration, measured with HIP events recorded on
                the launch stream by the library (fi_prof_*) during the timed steps.
  cpu_baseline  the CPU oracle (oracle/, kind "port") timed on this host for the hot-path
                OPERATORS of one step -- see cpu_baseline() for the exact sample.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
BF16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA (the 5 PF headline figure includes 2:1 sparsity)
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = fp32 vector peak


def crop_algorithmic_bytes(log_entry):
    """B_min of one pyramid RoIAlign launch: 4*N*C*ch*cw (write) + 4*C*sum_r U_r (U_r = distinct
    (row, col) taps of RoI r on its level) + 24*N (box, image index, level records)."""
    from feature_intertwiner_amd import _lib
    L = _lib.load()
    boxes, level, shapes, crop, C = (log_entry[k] for k in ("boxes", "level", "shapes", "crop", "depth"))
    N = boxes.size(0)
    dev = boxes.device
    total_u = 0
    for li, (H, W) in enumerate(shapes):
        sel = torch.nonzero(level == li + 2).view(-1)
        n = sel.numel()
        if n == 0:
            continue
        b = boxes[sel].contiguous()
        o = {k: torch.empty((n, crop), device=dev, dtype=torch.float32 if k.endswith("frac") else torch.int32)
             for k in ("y_valid", "y0", "y1", "y_frac", "x_valid", "x0", "x1", "x_frac")}
        _lib.check(L.fi_crop_and_resize_taps(_lib.ptr(b), n, H, W, crop, crop, _lib.ptr(o["y_valid"]),
                                             _lib.ptr(o["y0"]), _lib.ptr(o["y1"]), _lib.ptr(o["y_frac"]),
                                             _lib.ptr(o["x_valid"]), _lib.ptr(o["x0"]), _lib.ptr(o["x1"]),
                                             _lib.ptr(o["x_frac"]), _lib.current_stream()), "taps")

        def distinct(i0, i1, valid):
            v = torch.cat([torch.where(valid > 0, i0, torch.full_like(i0, -1)),
                           torch.where(valid > 0, i1, torch.full_like(i1, -1))], 1)
            v = torch.sort(v, dim=1)[0]
            new = torch.cat([torch.ones_like(v[:, :1], dtype=torch.bool), v[:, 1:] != v[:, :-1]], 1)
            return (new & (v >= 0)).sum(1)

        total_u += int((distinct(o["y0"], o["y1"], o["y_valid"]) * distinct(o["x0"], o["x1"], o["x_valid"])).sum())
    return 4 * N * C * crop * crop + 4 * C * total_u + 24 * N


def crop_dedup_bytes(log_entry, batch):
    """The same launch with every map element counted ONCE however many RoIs tap it: 4*N*C*ch*cw (write)
    + 4*C*|union over RoIs of their taps| per (image, level) + 24*N.  Training RoIs are jittered copies of a
    few objects and overlap heavily, so this is what can at most come from memory; B_min (per-RoI unique
    taps) is what the kernel must gather."""
    from feature_intertwiner_amd import _lib
    L = _lib.load()
    boxes, level, ind, shapes, crop, C = (log_entry[k] for k in ("boxes", "level", "box_ind", "shapes", "crop", "depth"))
    N = boxes.size(0)
    dev = boxes.device
    uniq = 0
    for li, (H, W) in enumerate(shapes):
        sel = torch.nonzero(level == li + 2).view(-1)
        n = sel.numel()
        if n == 0:
            continue
        b = boxes[sel].contiguous()
        o = {k: torch.empty((n, crop), device=dev, dtype=torch.float32 if k.endswith("frac") else torch.int32)
             for k in ("y_valid", "y0", "y1", "y_frac", "x_valid", "x0", "x1", "x_frac")}
        _lib.check(L.fi_crop_and_resize_taps(_lib.ptr(b), n, H, W, crop, crop, _lib.ptr(o["y_valid"]),
                                             _lib.ptr(o["y0"]), _lib.ptr(o["y1"]), _lib.ptr(o["y_frac"]),
                                             _lib.ptr(o["x_valid"]), _lib.ptr(o["x0"]), _lib.ptr(o["x1"]),
                                             _lib.ptr(o["x_frac"]), _lib.current_stream()), "taps")
        rows = torch.cat([o["y0"], o["y1"]], 1).long()                       # [n, 2*crop]
        rv = torch.cat([o["y_valid"], o["y_valid"]], 1) > 0
        cols = torch.cat([o["x0"], o["x1"]], 1).long()
        cv = torch.cat([o["x_valid"], o["x_valid"]], 1) > 0
        img = ind[sel].long().view(n, 1, 1)
        flat = (img * H + rows.unsqueeze(2)) * W + cols.unsqueeze(1)          # [n, 2c, 2c]
        ok = rv.unsqueeze(2) & cv.unsqueeze(1)
        mask = torch.zeros(batch * H * W, device=dev, dtype=torch.bool)
        mask[flat[ok]] = True
        uniq += int(mask.sum())
    return 4 * N * C * crop * crop + 4 * C * uniq + 24 * N


def north_star_roialign(dev, counters_only=False):
    """The north star's RoIAlign shape -- 512 RoIs x 256 channels x 7 x 7 on ONE map ([2, 256, 256, 256], jittered-GT +
    background RoIs of 4-128 pixels, seeded) -- through the reference-shaped operator on a channels-last map and on
    the reference's native NCHW map: kernel time from the in-library HIP events (50 launches each), priced with
    B_min (output + each RoI's distinct taps).  Outside the timed region; ~20 ms."""
    import numpy as np
    from feature_intertwiner_amd import _lib
    from feature_intertwiner_amd.roi_align.crop_and_resize import CropAndResizeFunction
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import training_rois
    rs = np.random.RandomState(2000)
    B, C, S, crop = 2, 256, 256, 7
    g = torch.Generator(device=dev).manual_seed(5)
    image = torch.randn(B, C, S, S, device=dev, generator=g)
    rois = torch.from_numpy(training_rois(rs, B, 256).reshape(-1, 4)).to(dev)
    ind = torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(256)
    N = rois.shape[0]
    entry = {"boxes": rois, "level": torch.full((N,), 2, device=dev, dtype=torch.int32), "shapes": [(S, S)],
             "crop": crop, "depth": C}
    b_min = crop_algorithmic_bytes(entry)
    fn = CropAndResizeFunction(crop, crop)
    out = {"shape": [N, C, crop, crop], "map": [B, C, S, S], "B_min_bytes": int(b_min), "peak": HBM_PEAK_GBPS, "unit": "GB/s"}
    if counters_only:       # the rocprofv3 --pmc child: the same launches, nothing timed
        with torch.no_grad():
            for img in (image.contiguous(memory_format=torch.channels_last), image):
                for _ in range(12):
                    fn(img, rois, ind)
        torch.cuda.synchronize()
        return None
    with torch.no_grad():
        for name, img, key in (("channels_last", image.contiguous(memory_format=torch.channels_last), "crop_fwd_nhwc_7x7"),
                               ("nchw", image, "crop_fwd_7x7")):
            for _ in range(10):
                fn(img, rois, ind)
            torch.cuda.synchronize()
            # five batches of ten launches, the MEDIAN batch average: one launch that pays for something else (a page
            # mapping after another process used the GPU: 3 ms once in 50 launches, seen in one evidence run) must not
            # set the number
            per_batch, n = [], 0
            for _ in range(5):
                _lib.prof_reset()
                _lib.prof_enable(True)
                for _ in range(10):
                    fn(img, rois, ind)
                torch.cuda.synchronize()
                _lib.prof_enable(False)
                nb, ms = _lib.prof_get(key)
                if nb:
                    per_batch.append(ms / nb * 1e3)
                    n += nb
            if n:
                us = sorted(per_batch)[len(per_batch) // 2]
                out[name] = {"kernel": _lib.kernel_name(key), "avg_launch_us": round(us, 2), "launches_timed": n,
                             "batch_avgs_us": [round(v, 2) for v in per_batch],
                             "achieved": round(b_min / (us * 1e-6) / 1e9, 1),
                             "frac": round(b_min / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)}
    _lib.prof_reset()
    return out


def configs4_slice(dev, steps=6, warmup=3):
    """The single-GPU slice of BASELINE configs[4] (`python bench.py --config cfg5`: ResNet-101-FPN, 1333 x 800 padded to
    1344^2, 2 images per GPU, 1000 RoIs per image + mask head, 16-bit MFMA convolutions) for a few steps after the
    headline's timed region, so that the driver's record carries a number for it."""
    from feature_intertwiner_amd import conv as ficonv
    from feature_intertwiner_amd.config import make_config
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    from feature_intertwiner_amd.workflow import set_optimizer, train_step
    torch.manual_seed(2000)
    cfg = make_config("resnet101", 1344, 2, 1000, dev_switch=True, loss_choice="ot", ot_L=50, conv_precision="bf16")
    model = MaskRCNN(cfg).to(dev)
    opt = set_optimizer(model, cfg.TRAIN)
    batch = synthetic_batch(2, 1344, device=dev, seed=2000)
    model.external_proposals = SyntheticProposals(batch[2], 1344, seed=7, cycle=16)
    model.generator = torch.Generator(device=dev).manual_seed(11)
    for _ in range(warmup):
        terms = train_step(model, opt, list(batch))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        terms = train_step(model, opt, list(batch))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    out = {"workload": "single-GPU slice of BASELINE configs[4]: resnet101-FPN, 1344x1344 (1333x800 padded, SURVEY Q8), "
                       "2 images/GPU, 1000 RoIs/image + mask head, OT intertwiner on, full train step",
           "dtype": "bf16 conv operands, fp32 accumulation, fp32 elsewhere", "steps": steps, "warmup": warmup,
           "ms_per_step": round(ms, 3), "value": round(2 * 1e3 / ms, 4), "unit": "images/sec",
           "finite_losses": bool(all(torch.isfinite(v) for v in terms.values())),
           "command": "python bench.py --config cfg5 (its own roofline / conv_stack objects: profiles/)"}
    del model, opt, batch
    ficonv.invalidate_step_state()
    torch.cuda.empty_cache()
    return out


def pmc_traffic(kernel_substrings, extra_args, timeout_s=170, child_flag="--pmc-child"):
    """HBM-side bytes per launch of the named kernels from rocprofv3's FETCH_SIZE / WRITE_SIZE, collected as
    MI355X_MICROARCH.md prescribes: each counter in its OWN `--pmc` pass (with --kernel-trace only), values in
    KB, and calibrated on a streaming copy of known size run in the same process (fi_calib_copy, 16 bytes
    per lane: on gfx950 FETCH_SIZE reports half the bytes of such a read).  The profiled child is this
    script with --pmc-child: the calibration copies, then the same train step.  Returns
    {substring: {"fetch": B, "write": B, "launches": n}}, "calibration": {...}} or {"error": ...}."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    out = {}
    calib_bytes = 256 * 1024 * 1024
    raw = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="fi_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
        cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--",
               sys.executable, os.path.abspath(__file__), child_flag] + extra_args
        try:
            subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=False,
                           cwd=os.environ.get("TMPDIR", "/tmp"))
        except subprocess.TimeoutExpired:
            shutil.rmtree(d, ignore_errors=True)
            return {"error": "rocprofv3 --pmc %s timed out after %d s" % (counter, timeout_s)}
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            shutil.rmtree(d, ignore_errors=True)
            return {"error": "rocprofv3 --pmc %s produced no counter file" % counter}
        per_dispatch = {}
        with open(files[0]) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") != counter:
                    continue
                key = (r.get("Dispatch_Id"), r["Kernel_Name"])
                per_dispatch[key] = per_dispatch.get(key, 0.0) + float(r["Counter_Value"])
        shutil.rmtree(d, ignore_errors=True)
        for (_, name), v in per_dispatch.items():
            raw.setdefault((counter, name), []).append(v * 1024.0)            # KB -> bytes
    def mean_of(counter, sub):
        # the library's profile slot "conv_wgrad_kernel<BM, R, S>" covers two device kernels: the row-major
        # conv_wgrad_vec_kernel (nearly every launch) and the scalar conv_wgrad_kernel
        alts = (sub, sub.replace("conv_wgrad_kernel<", "conv_wgrad_vec_kernel<")) if "conv_wgrad_kernel<" in sub else (sub,)
        vals = [v for (c, n), vs in raw.items() if c == counter and any(a in n for a in alts) for v in vs]
        return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)
    cf, ncal = mean_of("FETCH_SIZE", "fi_calib_copy_kernel")
    cw, _ = mean_of("WRITE_SIZE", "fi_calib_copy_kernel")
    if not cf or not cw:
        return {"error": "calibration kernel not found in the counter output"}
    f_fac, w_fac = calib_bytes / cf, calib_bytes / cw
    out["calibration"] = {"kernel": "fi_calib_copy_kernel (256 MiB read + 256 MiB written, 16 B/lane)",
                          "FETCH_SIZE_reported_MB": round(cf / 1e6, 1), "WRITE_SIZE_reported_MB": round(cw / 1e6, 1),
                          "fetch_factor": round(f_fac, 3), "write_factor": round(w_fac, 3), "launches": ncal}
    for sub in kernel_substrings:
        fe, n = mean_of("FETCH_SIZE", sub)
        wr, _ = mean_of("WRITE_SIZE", sub)
        if fe is not None and wr is not None:
            out[sub] = {"fetch": fe * f_fac, "write": wr * w_fac, "launches": n}
    return out


def _pin_to_physical_cores(limit=128):
    """One hardware thread per physical core (at most `limit`): the CPU leg's OpenMP / oneDNN threads then never share a
    core, which is what made its number differ by +-40 % between boxes.  Returns the previous affinity mask (or None)."""
    try:
        prev = os.sched_getaffinity(0)
        seen, cpus = set(), []
        for c in sorted(prev):
            path = "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c
            sib = open(path).read().strip() if os.path.exists(path) else str(c)
            if sib not in seen:
                seen.add(sib)
                cpus.append(c)
        cpus = cpus[:limit]
        if cpus:
            os.sched_setaffinity(0, cpus)
            torch.set_num_threads(len(cpus))
        return prev
    except Exception:
        return None


def cpu_conv_stack_seconds(shape_log, top=5, budget_s=60.0):
    """The step's convolutions (forward + input/weight gradients) on the host CPU through torch /
    oneDNN -- which is how the reference runs them on its CPU path.  The `top` layer shapes that carry the most
    flops (multiplicity included) are timed at their FULL batch, forward + backward, TWICE -- the first run pays
    oneDNN's primitive creation and the page faults of fresh buffers and is discarded, the median of three more runs
    counts --; the remaining shapes are priced at the seconds per flop measured on those.  Stops early (and
    extrapolates more) after `budget_s`."""
    import collections
    import torch.nn.functional as F
    counts = collections.Counter(shape_log)
    flops = lambda k: 2.0 * k[0] * k[4] * k[1] * k[5] * k[6] * ((k[2] + 2 * k[8][0] - k[5]) // k[7][0] + 1) * \
        ((k[3] + 2 * k[8][1] - k[6]) // k[7][1] + 1)
    todo = sorted(counts.items(), key=lambda kv: -flops(kv[0]) * kv[1])
    total, timed_flops, timed_s, rest_flops = 0.0, 0.0, 0.0, 0.0
    t_start = time.time()
    n_timed = 0
    for k, cnt in todo:
        N, Cin, H, W, Cout, R, S, st, pd = k
        if n_timed >= top or time.time() - t_start > budget_s:
            rest_flops += 3.0 * flops(k) * cnt
            continue
        x = torch.randn(N, Cin, H, W, requires_grad=True)
        w = torch.randn(Cout, Cin, R, S, requires_grad=True)
        runs = []
        for _ in range(4):                 # the first run is discarded, the MEDIAN of the next three is the shape's time
            t = time.time()                # (round 4: one timed run; the driver's box and the builder's differed by 40 %)
            y = F.conv2d(x, w, None, st, pd)
            y.backward(torch.ones_like(y))
            runs.append(time.time() - t)
        dt = sorted(runs[1:])[1]
        n_timed += 1
        total += dt * cnt
        timed_s += dt * cnt
        timed_flops += 3.0 * flops(k) * cnt
    if rest_flops and timed_flops:
        total += rest_flops * (timed_s / timed_flops)
    return total, n_timed, (timed_flops / max(timed_flops + rest_flops, 1.0))


def cpu_baseline(model, batch, log_entries, shape_log=None, conv_top=5, conv_budget_s=60.0, nms_boxes=6000, image=1024,
                 sinkhorn_problems=240):
    """ONE step's hot path on the host CPU.  Operators on the CPU oracle (all host cores for RoIAlign
    forward, one thread for backward / NMS / Sinkhorn, as the reference's C and Python do): the
    step's RoIAlign 7x7 and 14x14 forward on the real RoIs and feature-map sizes, their backward,
    NMS of 4 x 6000 clustered proposals, and 240 Sinkhorn problems (256 samples, L=50).  Conv stack
    through torch/oneDNN on all cores (cpu_conv_stack_seconds), as the reference's CPU path does."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle as O
    from helpers import clustered_dets
    O.build()
    prev_aff = _pin_to_physical_cores()
    rs = np.random.RandomState(2000)
    t_total = 0.0
    detail = {}
    frac = None
    bs = batch[0].size(0)
    for e in log_entries:
        crop, C = e["crop"], e["depth"]
        boxes = e["boxes"].cpu().numpy()
        level = e["level"].cpu().numpy()
        ind = e["box_ind"].cpu().numpy()
        t_f = t_b = 0.0
        for li, (H, W) in enumerate(e["shapes"]):
            sel = np.nonzero(level == li + 2)[0]
            if len(sel) == 0:
                continue
            fmap = rs.standard_normal((bs, C, H, W)).astype(np.float32)
            t = time.time()
            out = O.crop_and_resize_forward(fmap, boxes[sel], ind[sel], crop, crop)
            t_f += time.time() - t
            t = time.time()
            O.crop_and_resize_backward(out, boxes[sel], ind[sel], fmap.shape)
            t_b += time.time() - t
        detail["roialign_%dx%d_fwd_ms" % (crop, crop)] = t_f * 1e3
        detail["roialign_%dx%d_bwd_ms" % (crop, crop)] = t_b * 1e3
        t_total += t_f + t_b
    dets = [clustered_dets(rs, nms_boxes, image) for _ in range(bs)]
    t = time.time()
    for d in dets:
        O.pth_nms(d, 0.7)
    detail["nms_ms"] = (time.time() - t) * 1e3
    t_total += time.time() - t
    x = np.maximum(rs.standard_normal((max(sinkhorn_problems, 1), 256, 1)), 0).astype(np.float32)
    y = np.maximum(rs.standard_normal((max(sinkhorn_problems, 1), 256, 1)), 0).astype(np.float32)
    t = time.time()
    for p in range(sinkhorn_problems):           # every problem of the step (~1 s; rounds 1-3 timed 24 and scaled)
        O.sinkhorn(x[p], y[p], 1.0, 50)
    sk = time.time() - t
    detail["sinkhorn_%d_ms" % sinkhorn_problems] = sk * 1e3
    t_total += sk
    sample = ("one step's operator work on the CPU oracle: RoIAlign 7x7+14x14 fwd (OpenMP, %d threads) "
              "and bwd (serial) on the step's %d RoIs, NMS %dx%d @0.7 (serial), Sinkhorn %dx256x256 "
              "L=50 (serial, all timed)" % (O.num_threads(), log_entries[0]["boxes"].size(0), bs, nms_boxes, sinkhorn_problems))
    unit = "images/sec (hot-path operators only, conv stack excluded)"
    if shape_log:
        conv_s, n_shapes, frac = cpu_conv_stack_seconds(shape_log, top=conv_top, budget_s=conv_budget_s)
        detail["conv_stack_fwd_bwd_ms"] = conv_s * 1e3
        detail["operators_ms"] = t_total * 1e3
        t_total += conv_s
        unit = "images/sec (operators on the CPU oracle + conv stack on torch CPU/oneDNN)"
        sample += ("; conv stack: the %d layer shapes with the most flops timed fwd+bwd at full batch with torch on %d "
                   "threads (second of two runs), = %.0f%% of the conv flops; the other shapes priced at the measured "
                   "seconds per flop" % (n_shapes, torch.get_num_threads(), 100 * frac))
    used = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else O.num_threads()
    if prev_aff is not None:
        try:
            os.sched_setaffinity(0, prev_aff)
        except Exception:
            pass
    return {"value": bs / t_total, "unit": unit,
            "cores": min(O.num_threads(), used), "kind": "port", "sample": sample,
            "conv_flops_timed_share": None if frac is None else round(frac, 3),
            "conv_flops_extrapolated_share": None if frac is None else round(1.0 - frac, 3),
            "affinity": "one hardware thread per physical core (sched_setaffinity), %d CPUs" % used,
            "detail_ms": {k: round(v, 1) for k, v in detail.items()}, "host_cpus": os.cpu_count()}


def cpu_baseline_configs0(dev):
    """BASELINE configs[0] -- ResNet-50-FPN, 2 synthetic 512 x 512 images, 64 RoIs per image, OT off: the reference's own
    CPU-runnable case -- as a CPU composite: one step of that model is run on the GPU once to log its RoIAlign launches and
    convolution shapes, then the operators are timed on the CPU oracle and EVERY convolution shape (forward + backward)
    on torch / oneDNN (the model is small enough that nothing is extrapolated)."""
    from feature_intertwiner_amd import conv as ficonv
    from feature_intertwiner_amd.config import make_config
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.roi_align import crop_and_resize as car
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    from feature_intertwiner_amd.workflow import set_optimizer, train_step
    torch.manual_seed(2000)
    cfg = make_config("resnet50", 512, 2, 64, dev_switch=False, loss_choice="l2")
    model = MaskRCNN(cfg).to(dev)
    opt = set_optimizer(model, cfg.TRAIN)
    batch = synthetic_batch(2, 512, device=dev, seed=2000)
    model.external_proposals = SyntheticProposals(batch[2], 512, seed=7)
    model.generator = torch.Generator(device=dev).manual_seed(11)
    for _ in range(2):
        train_step(model, opt, list(batch), do_meta=False)
    car.LAUNCH_LOG, ficonv.SHAPE_LOG = [], []
    train_step(model, opt, list(batch), do_meta=False)
    torch.cuda.synchronize(dev)
    log, shapes = car.LAUNCH_LOG, ficonv.SHAPE_LOG
    car.LAUNCH_LOG = ficonv.SHAPE_LOG = None
    entries = [e for e in log if e["crop"] in (7, 14) and e["boxes"].size(0) == 2 * 64][-2:]
    if not entries:
        entries = [e for e in log if e["crop"] in (7, 14)][-2:]
    out = cpu_baseline(model, batch, entries, shapes, conv_top=10 ** 6, conv_budget_s=120.0, nms_boxes=6000, image=512,
                       sinkhorn_problems=0)
    out["workload"] = "BASELINE configs[0]: ResNet-50-FPN, 2 x 512^2, 64 RoIs/image, OT off (no Sinkhorn problems)"
    del model, opt
    torch.cuda.empty_cache()
    return out


def issue_profile(step, dev, ms_per_step_unprofiled, steps=2):
    """What the HOST side of a step costs and how busy the GPU is (at 8 ranks x 2 images the device step halves and
    the launch path is the first thing that can cap scaling).
      host_issue_ms_per_step  wall time of step() when the device queue is EMPTY at its start (synchronise, then time the
                              call): what the host needs to issue one step, never blocked on a full queue;
      gpu_busy_ms_per_step    union over all streams of the kernel / copy intervals of `steps` traced steps (torch.profiler);
      gpu_idle_ms_per_step    ms_per_step of the UN-profiled timed region minus that union (the tracer slows the host, so
                              gaps measured inside a traced run over-state the idle time; kernel durations are not affected);
      launches / framework_launches  device kernels per step, and those that are not this library's (torch glue)."""
    from torch.profiler import ProfilerActivity, profile
    host = []
    for _ in range(steps):
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        step()
        host.append(time.perf_counter() - t)
    torch.cuda.synchronize(dev)
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(steps):
            step()
        torch.cuda.synchronize(dev)
    iv, launches, fw = [], 0, 0
    for e in prof.events():
        if e.device_type != torch.autograd.DeviceType.CUDA:
            continue
        iv.append((e.time_range.start, e.time_range.end))
        name = e.name.lower()
        if "memcpy" in name or "memset" in name:
            continue
        launches += 1
        # this library's kernels live in anonymous namespaces of csrc/*.hip (demangled "(anonymous namespace)::...") or are
        # named fi_*; everything else on the device is torch glue (at::native::..., rocprim, copies as kernels) or RCCL
        ours = ("(anonymous namespace)::" in e.name and "at::" not in e.name) or name.startswith("fi_") or "fi_calib" in name
        if not ours and "nccl" not in name and "rccl" not in name:
            fw += 1
    iv.sort()
    busy, cs, ce = 0.0, None, None
    for a, b in iv:
        if ce is None:
            cs, ce = a, b
        elif a <= ce:
            ce = max(ce, b)
        else:
            busy += ce - cs
            cs, ce = a, b
    if ce is not None:
        busy += ce - cs
    busy_ms = busy / steps / 1e3
    return {"host_issue_ms_per_step": round(min(host) * 1e3, 2), "gpu_busy_ms_per_step": round(busy_ms, 2),
            "gpu_idle_ms_per_step": round(max(ms_per_step_unprofiled - busy_ms, 0.0), 2),
            "launches_per_step": launches // steps, "framework_launches_per_step": fw // steps,
            "method": "host: step() timed from an empty device queue; busy: union of kernel intervals over all streams of %d "
                      "traced steps (torch.profiler); idle = un-profiled ms_per_step - busy" % steps}


def graph_ab_child():
    """`python bench.py --batch-per-gpu 2 --graph-ab-child` in a process of its own; its JSON line."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--batch-per-gpu", "2", "--graph-ab-child", "--no-cpu-baseline",
           "--no-pmc", "--no-dense-reference", "--steps", "8", "--warmup", "3"]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=400)
        lines = [l for l in r.stdout.decode("utf-8", "replace").splitlines() if l.startswith("{")]
        return json.loads(lines[-1]) if lines else {"error": "no output (exit code %d)" % r.returncode}
    except Exception as ex:
        return {"error": repr(ex)}


def graph_ab_two_images(model, opt, dev, image_size, train_step, steps=8, own_model=False, eager=None):
    """BASELINE configs[3]'s per-GPU shape (2 images) on this GPU, eager against ONE hipGraph replay of the whole step
    (possible because the step has no host synchronisation): ms/step and GPU idle time of both.  At 2 images the device step
    is ~60 ms against ~25 ms of host issue time; the replay takes the host out of the picture entirely."""
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    keep_ext, keep_gen = model.external_proposals, model.generator
    batch2 = synthetic_batch(2, image_size, device=dev, seed=4000)
    model.external_proposals = SyntheticProposals(batch2[2], image_size, seed=9, cycle=16)
    model.generator = torch.Generator(device=dev).manual_seed(13)

    def step2():
        return train_step(model, opt, list(batch2), do_meta=True)

    def timed(fn):
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t) / steps * 1e3
    out = {}
    try:
        if eager is not None:               # the caller's timed region IS the eager leg (a 2-image process)
            out["eager"] = eager
            for _ in range(2):
                step2()
        else:
            for _ in range(3):              # (the caching allocator's pools are per stream: warm the one that is timed)
                step2()
            ms_e = timed(step2)
            out["eager"] = dict(ms_per_step=round(ms_e, 3), **issue_profile(step2, dev, ms_e))
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(3):
                step2()
        torch.cuda.current_stream(dev).wait_stream(s)
        g = torch.cuda.CUDAGraph()
        for gen in (model.generator, model.external_proposals.gen):
            g.register_generator_state(gen)
        with torch.cuda.graph(g, stream=s):
            step2()
        torch.cuda.synchronize(dev)
        timed(g.replay)
        ms_g = timed(g.replay)
        host = []
        for _ in range(3):
            torch.cuda.synchronize(dev)
            t = time.perf_counter()
            g.replay()
            host.append(time.perf_counter() - t)
        torch.cuda.synchronize(dev)
        # (no busy / idle figures for the replay: the tracer does not see every kernel node of a graph launch)
        out["graph_replay"] = {"ms_per_step": round(ms_g, 3), "host_issue_ms_per_step": round(min(host) * 1e3, 2)}
        del g
    except Exception as ex:           # a capture failure must not take the headline line with it
        out["error"] = repr(ex)
    model.external_proposals, model.generator = keep_ext, keep_gen
    out["what"] = ("2 images per GPU (BASELINE configs[3]'s per-rank shape) on one GPU, in a process of its own "
                   "(python bench.py --batch-per-gpu 2 --graph-ab-child): eager launches vs one hipGraph replay of the whole step")
    return out


def _self_launch(n):
    """Re-run this command line under `python -m torch.distributed.run --nnodes=1 --nproc-per-node n` and
    return the child job's JSON line."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    share = os.environ.get("FI_BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count()
    if have < n and not share:
        raise SystemExit("--gpus %d, but only %d GPU(s) are visible" % (n, have))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: needed by RCCL between processes on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)          # stderr passes through
    lines = [l for l in r.stdout.decode("utf-8", "replace").splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        sys.stderr.write(r.stdout.decode("utf-8", "replace"))
        raise SystemExit("the %d-rank job failed (exit code %d)" % (n, r.returncode))
    return lines[-1]


def _preflight(rank, world, dev, share):
    """Fail with a readable message BEFORE the model is built if the job cannot be what the JSON line will claim: every
    rank on its own physical GPU (an all-gather of device UUIDs over the store, no GPU traffic) and a 1-element
    all-reduce through the backend that will carry the gradients (RCCL communicator set-up, xGMI / dmabuf IPC)."""
    props = torch.cuda.get_device_properties(dev)
    ids = [None] * world
    dist.all_gather_object(ids, (rank, str(getattr(props, "uuid", "")) or "device-%d" % torch.cuda.current_device()))
    distinct = len(set(u for _, u in ids))
    if distinct < world and not share:
        raise SystemExit("bench.py preflight: %d ranks but only %d distinct GPUs (%s) -- one process per GPU is required; "
                         "check --nproc-per-node, LOCAL_RANK and HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES"
                         % (world, distinct, sorted(set(u for _, u in ids))))
    try:
        t = torch.ones(1, device=dev)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        got = float(t.item())
    except Exception as ex:
        raise SystemExit("bench.py preflight: the first all-reduce over backend '%s' failed on rank %d: %r "
                         "(HSA_ENABLE_IPC_MODE_LEGACY=%s; RCCL needs dmabuf IPC on this driver: export "
                         "HSA_ENABLE_IPC_MODE_LEGACY=0)" % (dist.get_backend(), rank, ex,
                                                            os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")))
    if got != float(world):
        raise SystemExit("bench.py preflight: all-reduce of ones over %d ranks returned %r" % (world, got))


def _overlap_report(profile, steps):
    """GradientBuckets.profile (HIP events on the main and the communication stream) -> how much of the gradient
    exchange ran inside the backward window.  Times in ms, averaged per step."""
    per_step, cur = [], None
    for rec in profile:
        if rec[0] == "backward_start":
            cur = {"start": rec[1], "buckets": []}
            per_step.append(cur)
        elif cur is not None and rec[0] == "bucket":
            cur["buckets"].append(rec[1:])
        elif cur is not None and rec[0] in ("backward_end", "reduced"):
            cur[rec[0]] = rec[1]
    per_step = [st for st in per_step if "backward_end" in st and "reduced" in st and st["buckets"]]
    if not per_step:
        return None
    acc = {"backward_ms": 0.0, "comm_busy_ms": 0.0, "comm_inside_backward_ms": 0.0, "exposed_after_backward_ms": 0.0,
           "first_bucket_at_ms": 0.0}
    for st in per_step:
        bwd = st["start"].elapsed_time(st["backward_end"])
        acc["backward_ms"] += bwd
        acc["exposed_after_backward_ms"] += max(0.0, st["backward_end"].elapsed_time(st["reduced"]))
        acc["first_bucket_at_ms"] += st["start"].elapsed_time(st["buckets"][0][2])
        for bi, nbytes, e0, e1 in st["buckets"]:
            t0, t1 = st["start"].elapsed_time(e0), st["start"].elapsed_time(e1)
            acc["comm_busy_ms"] += t1 - t0
            acc["comm_inside_backward_ms"] += max(0.0, min(t1, bwd) - min(t0, bwd))
    n = float(len(per_step))
    out = {k: round(v / n, 3) for k, v in acc.items()}
    out["steps"] = len(per_step)
    out["buckets_per_step"] = len(per_step[0]["buckets"])
    out["bytes_per_step"] = int(sum(b[1] for b in per_step[0]["buckets"]))
    out["frac_inside_backward"] = round(out["comm_inside_backward_ms"] / max(out["comm_busy_ms"], 1e-9), 4)
    out["method"] = ("HIP events: backward_start / backward_end on the compute stream, one pair around every bucket's "
                     "in-place RCCL all-reduce on the communication stream (the collective's busy time includes "
                     "waiting for the slowest rank); exposed = compute stream idle from the end of backward until "
                     "the last bucket is reduced")
    return out


def main():
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner to the C-level stdout when the
    # first communicator is created, and libraries may print more: everything written to fd 1 while the
    # benchmark runs is sent to stderr, and the JSON line goes to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        line = _main()
    finally:
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)        # C stdio buffers (the RCCL banner) must drain to stderr too
        except Exception:
            pass
        os.dup2(real_stdout, 1)
        os.close(real_stdout)
    if line is not None:
        print(line, flush=True)


def _main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--backbone", default="resnet101")
    ap.add_argument("--image-size", type=int, default=1024)
    ap.add_argument("--batch-per-gpu", type=int, default=4)
    ap.add_argument("--rois", type=int, default=512)
    ap.add_argument("--ot-L", type=int, default=50)
    ap.add_argument("--config", default="cfg3", choices=["cfg3", "cfg5"],
                    help="cfg3 = BASELINE configs[2], the headline (default); cfg5 = the single-GPU slice of BASELINE "
                         "configs[4]: 1333x800 padded to 1344^2, 2 images/GPU, 1000 RoIs/image + mask head, bf16 MFMA convs")
    ap.add_argument("--conv-precision", default=None, choices=["fp32", "bf16", "fp16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense-reference", action="store_true",
                    help="skip the 8 extra steps (after the timed region) that time the dense form of the backward pass")
    ap.add_argument("--dense-backward", action="store_true",
                    help="the backward pass in its dense (round-2) form: dense RPN and mask-head gradients over the "
                         "anchors / RoIs whose gradient is identically zero, one elementwise pass per BatchNorm layer, "
                         "autograd accumulating every multi-reader gradient (conv.GATES = conv._UNSCALED_BACKWARD = False)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc passes that fill roofline.traffic")
    ap.add_argument("--preflight", action="store_true",
                    help="dry run of a --gpus N job: process group + first all-reduce, model and gradient buckets built, "
                         "NO training step; prints GPU_MAX_HW_QUEUES, the measured stream concurrency map and the bucket "
                         "schedule as one JSON line")
    ap.add_argument("--profile-steps", type=int, default=4,
                    help="extra steps AFTER the timed region, run with in-library HIP-event timing for the roofline objects")
    ap.add_argument("--no-issue-profile", action="store_true",
                    help="skip the host-issue / GPU-busy pass and the 2-image eager-vs-hipGraph A/B (runs traced from outside)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-child-roi", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--graph-ab-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--mask-head-on-positive-slots", action="store_true",
                    help="NOT the headline configuration: run the mask head only on the RoI slots that can hold "
                         "positives (identical loss/gradients, see MaskRCNN.forward); recorded in config.variant")
    args = ap.parse_args()
    if args.config == "cfg5":
        args.image_size, args.batch_per_gpu, args.rois = 1344, 2, 1000
        args.conv_precision = args.conv_precision or "bf16"
    args.conv_precision = args.conv_precision or "fp32"

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, exactly as the documented
            # command does (one process per GPU under torch.distributed.run on 127.0.0.1); rank 0 of the child job
            # prints the JSON line, which is passed through
            return _self_launch(args.gpus)
        raise SystemExit("--gpus %d does not match WORLD_SIZE=%d of the launcher" % (args.gpus, world))
    # FI_BENCH_SHARE_GPU=1 (testing only): all ranks share cuda:0 over the gloo backend, to exercise
    # the multi-process code path on a single-GPU box; the real launch is one rank per GPU over RCCL
    share = os.environ.get("FI_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # FI_DP_FORCE=1 (measurement only): run the data-parallel engine over RCCL even with ONE rank, to price its
    # per-step machinery (bucket packing, collectives, write-back) on a one-GPU box
    force_dp = os.environ.get("FI_DP_FORCE") == "1"
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if world > 1:
            _preflight(rank, world, dev, share)

    from feature_intertwiner_amd import _lib
    from feature_intertwiner_amd.config import make_config
    from feature_intertwiner_amd.data_parallel import GradientBuckets, all_reduce_statistics, broadcast_parameters
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.roi_align import crop_and_resize as car
    from feature_intertwiner_amd import conv as ficonv
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    from feature_intertwiner_amd.workflow import set_optimizer, train_step
    _lib.load()

    torch.manual_seed(2000)
    cfg = make_config(args.backbone, args.image_size, args.batch_per_gpu, args.rois, dev_switch=True,
                      loss_choice="ot", ot_L=args.ot_L, gpu_count=world, conv_precision=args.conv_precision)
    cfg.MRCNN.MASK_HEAD_ON_POSITIVE_SLOTS = bool(args.mask_head_on_positive_slots)
    if args.dense_backward:
        ficonv.GATES = ficonv._UNSCALED_BACKWARD = False
    model = MaskRCNN(cfg).to(dev)
    broadcast_parameters(model)
    opt = set_optimizer(model, cfg.TRAIN)
    sync = GradientBuckets(model) if (world > 1 or force_dp) else None
    reduce_fn = all_reduce_statistics if (world > 1 or force_dp) else None
    batch = synthetic_batch(args.batch_per_gpu, args.image_size, device=dev, seed=2000 + rank)
    model.external_proposals = SyntheticProposals(batch[2], args.image_size, seed=7 + rank, cycle=16)
    model.generator = torch.Generator(device=dev).manual_seed(11 + rank)

    def step():
        return train_step(model, opt, list(batch), do_meta=True, grad_sync=sync, world_size=world,
                          reduce_fn=reduce_fn)

    result_line = None
    if args.preflight:
        with torch.cuda.device(dev):
            _lib.side_stream(dev), _lib.side_stream3(dev)
            if sync is not None:
                sync.comm_stream
            rep = _lib.stream_report(dev)
        info = {"preflight": True, "rank": rank, "world": world, "backend": dist.get_backend() if dist.is_initialized() else None,
                "device": torch.cuda.get_device_name(dev), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                "GPU_MAX_HW_QUEUES": rep["GPU_MAX_HW_QUEUES"],
                "streams": [{k: v for k, v in r.items()} for r in rep["streams"]],
                "buckets": None if sync is None else [
                    {"bytes": 4 * (b.end - b.start), "parameters": len(b.params), "waits_for": len(b.wait)}
                    for b in sync.layout.buckets],
                "gradient_bytes": None if sync is None else 4 * sync.layout.total,
                "issue_order": "bucket 0 first (the last layers' gradients: autograd produces them first); a bucket leaves "
                               "from the hook of its last awaited gradient, on the communication stream"}
        if world > 1:
            allr = [None] * world
            dist.all_gather_object(allr, info)
            info = {"preflight": True, "ranks": allr}
        if world > 1 or force_dp:
            dist.destroy_process_group()
        return json.dumps(info) if rank == 0 else None
    if args.pmc_child_roi:
        # profiled under `rocprofv3 --pmc <one counter>`: calibration copies, then the north star's RoIAlign launches
        a = torch.empty(64 * 1024 * 1024, device=dev)
        b = torch.empty_like(a)
        for _ in range(4):
            _lib.check(_lib.load().fi_calib_copy(_lib.ptr(a), _lib.ptr(b), a.numel(), _lib.current_stream()), "calib")
        del a, b
        north_star_roialign(dev, counters_only=True)
        return None
    if args.pmc_child:
        # profiled under `rocprofv3 --pmc <one counter>`: calibration copies of known size, then the step
        a = torch.empty(64 * 1024 * 1024, device=dev)
        b = torch.empty_like(a)
        for _ in range(4):
            _lib.check(_lib.load().fi_calib_copy(_lib.ptr(a), _lib.ptr(b), a.numel(), _lib.current_stream()), "calib")
        del a, b
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        return None

    if sync is not None:
        # engine start-up, not part of the W warm-up steps: RCCL communicator + channel set-up happen at the first
        # collective, and GradientBuckets issues its buckets from autograd hooks (overlapped) only once it has seen
        # which parameters produce gradients for ABSENT_STEPS steps of this graph variant
        for _ in range(sync.ABSENT_STEPS + 1):
            step()
        # RCCL has opened its channels and streams by now: re-measure which picked streams still run next to each other
        # and replace those that do not (once, outside the warm-up and the timed region)
        torch.cuda.synchronize()
        sync.recheck_streams()
    for _ in range(args.warmup):
        terms = step()
    # ---- the timed region: exactly K steps, nothing recorded (no event timing, no logging) ----------
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        terms = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if sync is not None:
        sync.check()               # replicas stayed consistent (host sync, outside the timed region)
    # ---- host issue time / GPU busy and idle time / launches per step, this rank (outside the timed region) ----------
    issue = None
    if not args.no_issue_profile:
        try:
            issue = issue_profile(step, dev, elapsed / args.steps * 1e3)
        except Exception as ex:
            issue = {"error": repr(ex)}
    if args.graph_ab_child:
        # (a process of its own for the 2-image eager / hipGraph comparison: the same numbers inside the 4-image run came
        # out 6-15 ms slower for the eager leg than the dedicated `--batch-per-gpu 2` run -- allocator state of the other
        # workloads of that process)
        out = graph_ab_two_images(model, opt, dev, args.image_size, train_step, own_model=True,
                                  eager=dict(ms_per_step=round(elapsed / args.steps * 1e3, 3), **(issue or {})))
        return json.dumps(out)
    # ---- a separate profiled pass for the roofline objects (HIP events around every library kernel) --
    prof_steps = max(1, args.profile_steps)
    _lib.prof_reset()
    _lib.prof_enable(True)
    car.LAUNCH_LOG = []
    ficonv.FLOP_LOG = {}
    ficonv.SHAPE_LOG = []
    # EXCLUSIVE kernel durations: in the timed region the weight gradients run on a second stream next to the data
    # gradients (conv.WGRAD_SIDE_STREAM_MAX_PIXELS) and two kernels that share the chip each take longer than alone;
    # the roofline objects price a kernel against the whole chip, so this pass keeps everything on one stream
    # (`exclusive_ms_per_step` in the JSON is what that costs; `value` comes from the timed region above)
    side_pixels = ficonv.WGRAD_SIDE_STREAM_MAX_PIXELS
    ficonv.WGRAD_SIDE_STREAM_MAX_PIXELS = 0
    from feature_intertwiner_amd import model as fimodel
    dead_side = fimodel._DEAD_SIDE
    fimodel._DEAD_SIDE = False            # the mask head's unread batch back on the main stream, for the same reason
    from feature_intertwiner_amd import sub_module as fisub, workflow as fiwork
    keep_side = (fimodel._PROPOSAL_SIDE, fisub._BIG_SIDE, fiwork.META_SIDE_STREAM)
    fimodel._PROPOSAL_SIDE = fisub._BIG_SIDE = fiwork.META_SIDE_STREAM = False     # (round 4's side-stream work as well)
    t1 = time.perf_counter()
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize()
    prof_elapsed = time.perf_counter() - t1
    ficonv.WGRAD_SIDE_STREAM_MAX_PIXELS = side_pixels
    fimodel._DEAD_SIDE = dead_side
    fimodel._PROPOSAL_SIDE, fisub._BIG_SIDE, fiwork.META_SIDE_STREAM = keep_side
    _lib.prof_enable(False)
    log = car.LAUNCH_LOG
    car.LAUNCH_LOG = None
    flop_log = ficonv.FLOP_LOG
    ficonv.FLOP_LOG = None
    shape_log = ficonv.SHAPE_LOG
    ficonv.SHAPE_LOG = None
    shape_log = shape_log[:len(shape_log) // prof_steps]       # the convolutions of ONE step
    # ---- the same step with the backward pass in its dense form, for reference (outside the timed region) ----------
    dense_ref = None
    if world == 1 and not args.dense_backward and not args.no_dense_reference:
        keep = (ficonv.GATES, ficonv._UNSCALED_BACKWARD)
        ficonv.GATES = ficonv._UNSCALED_BACKWARD = False
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(6):
            step()
        torch.cuda.synchronize()
        dense_ms = (time.perf_counter() - t2) / 6 * 1e3
        ficonv.GATES, ficonv._UNSCALED_BACKWARD = keep
        step()                                  # back to the default form (plans, W^T tables)
        torch.cuda.synchronize()
        dense_ref = {"ms_per_step": round(dense_ms, 3), "value": round(args.batch_per_gpu * 1e3 / dense_ms, 4), "steps": 6,
                     "what": "python bench.py --dense-backward: dense RPN / mask-head gradients over anchors and RoIs whose "
                             "gradient is identically zero, one elementwise pass per BatchNorm layer, autograd accumulating "
                             "multi-reader gradients -- the same forward results and the same gradients as the default"}
    # ---- the default backward pass against its dense form on the SAME weights / inputs / draws (outside the timed region)
    backward_check = None
    if world == 1 and sync is None and not args.dense_backward and not args.no_dense_reference:
        # (not with the data-parallel engine attached: its autograd hooks expect begin() / sync() around a backward pass)
        from feature_intertwiner_amd.workflow import check_backward_forms
        lowp = args.conv_precision != "fp32"
        r = check_backward_forms(model, batch, bar=6e-2 if lowp else 2e-5,
                                 skip=(lambda n: n.startswith("ot_loss") or n.startswith("dev_roi.feat_extract")) if lowp else None)
        backward_check = {"max_rel_dev": float("%.3g" % r["max_rel_dev"]), "worst": r["worst"], "params": r["params"],
                          "loss_rel": float("%.3g" % r["loss_rel"]), "none_sets_equal": r["none_sets_equal"],
                          "attempts": r["attempts"], "boundary_events": [{"max_rel_dev": float("%.3g" % v["max_rel_dev"]),
                                                                          "evidence": v["evidence"]} for v in r["boundary_events"]],
                          "what": "one backward pass in the default form and one in the dense form "
                                  "(workflow.check_backward_forms) after the timed steps: max over parameters of "
                                  "max|g - g_dense| / max|g_dense|; boundary_events: a pass set aside because pre-activations of "
                                  "the RPN's shared convolution fell on the other side of their ReLU (the default form evaluates "
                                  "it at the sampled anchors as a matrix product -- another summation order): VERIFIED (evidence: "
                                  "at every differing channel the dense kernel and the row form disagree on the sign of a sampled "
                                  "pre-activation whose float64 value is within 16 x 2^-24 of its summands' magnitude) and then "
                                  "REPLAYED on the same draws with the row form using the dense kernel's mask bits; max_rel_dev "
                                  "is the replay's"}
        step()                                  # plans / W^T tables back in the default form
        torch.cuda.synchronize()
    # ---- BASELINE configs[4], single-GPU slice, on the driver's record too (outside the timed region) -----------------
    cfg4_slice = None
    if world == 1 and sync is None and args.config == "cfg3" and not args.no_dense_reference and not args.dense_backward and \
            args.conv_precision == "fp32":
        try:
            cfg4_slice = configs4_slice(dev)
        except Exception as ex:          # never takes the headline down
            cfg4_slice = {"error": repr(ex)}
    graph_ab = None
    if world == 1 and sync is None and args.config == "cfg3" and not args.no_dense_reference and not args.dense_backward and \
            args.conv_precision == "fp32" and args.batch_per_gpu != 2 and not args.no_issue_profile:
        graph_ab = graph_ab_child()
    per_rank_ms, rccl_ranks, overlap = None, None, None
    if sync is not None and not share:
        # a further pass with HIP events around every bucket's collective: how much of the exchange hides in backward
        sync.profile = []
        for _ in range(prof_steps):
            step()
        torch.cuda.synchronize()
        overlap = _overlap_report(sync.profile, prof_steps)
        sync.profile = None
    configs3 = None
    if world > 1 and args.config == "cfg3" and args.batch_per_gpu != 2:
        # BASELINE configs[3] quotes 2 images per GPU: the same model and engine on a 2-image shard, after the timed
        # region (2 warm-up + 6 timed steps, barrier + synchronize on both sides, MAX over ranks)
        batch2 = synthetic_batch(2, args.image_size, device=dev, seed=3000 + rank)
        keep_ext = model.external_proposals
        model.external_proposals = SyntheticProposals(batch2[2], args.image_size, seed=17 + rank, cycle=16)

        def step2():
            return train_step(model, opt, list(batch2), do_meta=True, grad_sync=sync, world_size=world, reduce_fn=reduce_fn)
        for _ in range(2):
            step2()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for _ in range(6):
            step2()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        mine2 = torch.tensor([time.perf_counter() - t3], device=dev, dtype=torch.float64)
        dist.all_reduce(mine2, op=dist.ReduceOp.MAX)
        ms2 = float(mine2.item()) / 6 * 1e3
        # how much of that step is NOT convolution time: 2 further steps with HIP events around every library kernel and
        # everything on one stream (as the roofline pass above): the MFMA kernels' exclusive time per step, this rank.
        # The difference to ms_per_step is what the latency kernels of the side streams, the RoI operators, the gradient
        # exchange and the launch gaps add on the critical path at half the headline's conv time.
        _lib.prof_reset()
        _lib.prof_enable(True)
        side_pixels2 = ficonv.WGRAD_SIDE_STREAM_MAX_PIXELS
        ficonv.WGRAD_SIDE_STREAM_MAX_PIXELS = 0
        for _ in range(2):
            step2()
        torch.cuda.synchronize()
        ficonv.WGRAD_SIDE_STREAM_MAX_PIXELS = side_pixels2
        _lib.prof_enable(False)
        conv_ms = side_ms = 0.0
        for k in _lib.KERNEL_IDS:
            n, ms = _lib.prof_get(k)
            if n and k.startswith("conv"):
                conv_ms += ms / 2
            elif n and (k.startswith("nms") or k.startswith("proposal") or k.startswith("sinkhorn")):
                side_ms += ms / 2
        try:
            issue2 = issue_profile(step2, dev, ms2)
        except Exception as ex:
            issue2 = {"error": repr(ex)}
        configs3 = {"workload": "BASELINE configs[3]: 2 images per GPU, otherwise as config.workload", "images_per_gpu": 2,
                    "rank0_issue": issue2,
                    "global_batch": 2 * world, "steps": 6, "ms_per_step": round(ms2, 3),
                    "value": round(2 * world * 1e3 / ms2, 4), "unit": "images/sec",
                    "rank0_conv_kernels_ms_per_step": round(conv_ms, 3),
                    "rank0_not_conv_ms_per_step": round(ms2 - conv_ms, 3),
                    "rank0_nms_proposal_sinkhorn_kernels_ms_per_step": round(side_ms, 3)}
        model.external_proposals = keep_ext
        del batch2
    if world > 1:
        mine = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [round(float(t.item()) / args.steps * 1e3, 3) for t in every]
        elapsed = max(float(t.item()) for t in every)                 # MAX over ranks
        # which physical GPU every rank ran on (an all-gather of device UUIDs): N distinct devices = N RCCL ranks
        ids = [None] * world
        props = torch.cuda.get_device_properties(dev)
        dist.all_gather_object(ids, {"rank": rank, "device": torch.cuda.current_device(), "name": props.name,
                                     "uuid": str(getattr(props, "uuid", "")), "pid": os.getpid()})
        rccl_ranks = {"backend": dist.get_backend(), "world_size": world, "ranks": ids,
                      "distinct_devices": len(set((d["uuid"] or d["device"]) for d in ids))}

    if rank == 0:
        global_batch = args.batch_per_gpu * world
        ms_per_step = elapsed / args.steps * 1e3
        value = global_batch * args.steps / elapsed
        # ---- roofline of the RoIAlign 7x7 forward kernel ---------------------------------
        # (the Dev stage's maps are channels-last: crop_fwd_cl_kernel<7, 7>; NCHW kernel otherwise)
        n7, ms7 = _lib.prof_get("crop_fwd_nhwc_7x7")
        fwd7 = [e for e in log if e["crop"] == 7 and e["pyramid"] and e.get("nhwc")]
        roi_kernel = "crop_fwd_nhwc_7x7"
        if not (n7 and fwd7):
            n7, ms7 = _lib.prof_get("crop_fwd_7x7")
            fwd7 = [e for e in log if e["crop"] == 7 and e["pyramid"] and not e.get("nhwc")]
            roi_kernel = "crop_fwd_7x7"
        roof_roi = None
        if n7 and fwd7:
            e = fwd7[-1]
            b_alg = crop_algorithmic_bytes(e)
            b_dedup = crop_dedup_bytes(e, args.batch_per_gpu)
            dur = ms7 / n7 * 1e-3
            ach = b_alg / dur / 1e9
            roof_roi = {"kernel": _lib.kernel_name(roi_kernel), "map_layout": "NHWC" if e.get("nhwc") else "NCHW",
                    "bound": "hbm", "achieved": round(ach, 1),
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
                    "traffic": None, "algorithmic_bytes_per_launch": int(b_alg),
                    "avg_launch_us": round(dur * 1e6, 2), "launches_timed": n7,
                    "rois_per_launch": int(e["boxes"].size(0)), "bytes_model": "B_min (SURVEY 8d): output + each RoI's distinct taps",
                    "dedup_bytes_per_launch": int(b_dedup),
                    "achieved_dedup": round(b_dedup / dur / 1e9, 1), "frac_dedup": round(b_dedup / dur / 1e9 / HBM_PEAK_GBPS, 4),
                    "dedup_model": "output + the UNION of all RoIs' taps per (image, level): the jittered-GT RoIs overlap, "
                                   "so this is the most that can come from memory; the rest of B_min is served by L2"}
        kern = {}
        for k in _lib.KERNEL_IDS:
            n, ms = _lib.prof_get(k)
            if n:
                kern[_lib.kernel_name(k)] = {"launches": n, "avg_us": round(ms / n * 1e3, 2),
                                             "ms_per_step": round(ms / prof_steps, 3), "_key": k}
        # ---- roofline of the DOMINANT kernel (largest share of the step) ------------------------
        dom = max(kern.items(), key=lambda kv: kv[1]["ms_per_step"]) if kern else None
        roof = None
        if dom is not None and dom[1]["_key"] in flop_log:
            launches, flops = flop_log[dom[1]["_key"]]
            n, ms = _lib.prof_get(dom[1]["_key"])
            ach = flops / (ms * 1e-3) / 1e12
            peak = BF16_MFMA_PEAK_TFLOPS if "bf16" in dom[0] else FP32_MFMA_PEAK_TFLOPS
            roof = {"kernel": dom[0], "bound": "mfma", "achieved": round(ach, 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                    "algorithmic_flops_per_launch": int(flops / max(launches, 1)),
                    "avg_launch_us": round(ms / n * 1e3, 2), "launches_timed": n,
                    "share_of_step": round(dom[1]["ms_per_step"] / (prof_elapsed / prof_steps * 1e3), 4),
                    "flops_model": "2*N*Cout*OH*OW*Cin*R*S per launch (%s)" % (
                        "bf16 operands, fp32 accumulation" if "bf16" in dom[0] else "fp32, exact MFMA")}
        elif roof_roi is not None:
            roof = roof_roi
        # ---- flop-weighted efficiency of the whole conv stack (every MFMA kernel instance) ----------
        conv_stack = None
        if flop_log:
            tot_f = tot_ms = 0.0
            per = {}
            for name, v in kern.items():
                if v["_key"] in flop_log:
                    launches, flops = flop_log[v["_key"]]
                    n, ms = _lib.prof_get(v["_key"])
                    tot_f += flops
                    tot_ms += ms
                    per[name] = {"tflops": round(flops / (ms * 1e-3) / 1e12, 1), "ms_per_step": v["ms_per_step"],
                                 "gflop_per_step": round(flops / prof_steps / 1e9, 1)}
            if tot_ms > 0:
                ach = tot_f / (tot_ms * 1e-3) / 1e12
                cs_peak = FP32_MFMA_PEAK_TFLOPS if args.conv_precision == "fp32" else BF16_MFMA_PEAK_TFLOPS
                conv_stack = {"achieved": round(ach, 2), "peak": cs_peak, "unit": "TFLOP/s",
                              "frac": round(ach / cs_peak, 4),
                              "tflop_per_step": round(tot_f / prof_steps / 1e12, 3),
                              "ms_per_step": round(tot_ms / prof_steps, 2), "per_kernel": per,
                              "note": "sum of algorithmic flops / sum of kernel time over every conv_fwd (forward + "
                                      "data gradient) and conv_wgrad launch of the timed steps"}
        # ---- NMS and Sinkhorn figures -------------------------------------------------------
        n_m, ms_m = _lib.prof_get("nms_mask")
        n_s, ms_s = _lib.prof_get("nms_scan")
        nms_obj = None
        if n_m:
            nb, nbx = args.batch_per_gpu, cfg.RPN.PRE_NMS_LIMIT
            b_nms = nb * (20 * nbx + 8 * nbx * ((nbx + 63) // 64))
            us_m, us_s = ms_m / n_m * 1e3, (ms_s / n_s * 1e3 if n_s else 0.0)
            nms_obj = {"images": nb, "boxes_per_image": nbx, "mask_kernel_us": round(us_m, 1), "scan_kernel_us": round(us_s, 1),
                       "algorithmic_bytes": b_nms, "mask_GBps": round(b_nms / (us_m * 1e-6) / 1e9, 1),
                       "mask_frac_hbm": round(b_nms / (us_m * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                       "pair_tests_per_us": round(nb * nbx * nbx / 2 / us_m, 0),
                       "bytes_model": "20*N read + 8*N*ceil(N/64) mask written, per image (SURVEY 8d); the kernel is "
                                      "bound by the N^2/2 IoU tests (VALU), not by these bytes"}
        n_k, ms_k = _lib.prof_get("sinkhorn")
        sk_obj = None
        if n_k:
            ncls = cfg.DATASET.NUM_CLASSES - 1
            fl = 3 * ncls * 2 * args.ot_L * 2 * 256 * 256
            us_k = ms_k / n_k * 1e3
            sk_obj = {"problems": 3 * ncls, "samples": 256, "L": args.ot_L, "kernel_us": round(us_k, 1),
                      "GFLOPs": round(fl / (us_k * 1e-6) / 1e9, 1),
                      "frac_fp32_vector_peak": round(fl / (us_k * 1e-6) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                      "bound": "on-chip (fp32 FMA + LDS/barrier latency); the HBM roofline does not apply: "
                               "inputs are 2 x 256 floats per problem"}
        # ---- HBM-side traffic of the dominant kernel and of RoIAlign from the PMC counters ----------------
        if world == 1 and not args.no_pmc and roof is not None:
            child = ["--backbone", args.backbone, "--image-size", str(args.image_size), "--batch-per-gpu",
                     str(args.batch_per_gpu), "--rois", str(args.rois), "--ot-L", str(args.ot_L), "--conv-precision",
                     args.conv_precision]
            if args.mask_head_on_positive_slots:
                child.append("--mask-head-on-positive-slots")
            if args.dense_backward:
                child.append("--dense-backward")
            # kernel-name prefix the trace rows are matched on: "name<args" without the closing bracket for a template
            # instantiation, the plain name otherwise (the 16-bit kernels' slots)
            kn = roof["kernel"]
            subs = [kn.split("<")[0] + "<" + kn.split("<")[1].rstrip(">") if "<" in kn else kn]
            if roof_roi is not None:
                subs.append(roof_roi["kernel"].rstrip(">"))
            t_p = time.time()
            tr = pmc_traffic(subs, child)
            if "error" in tr:
                roof["traffic_error"] = tr["error"]
            else:
                for obj, sub in ((roof, subs[0]), (roof_roi, subs[1] if len(subs) > 1 else None)):
                    if obj is not None and sub in tr:
                        obj["traffic"] = int(tr[sub]["fetch"] + tr[sub]["write"])
                        obj["traffic_detail"] = {"fetch_bytes": int(tr[sub]["fetch"]), "write_bytes": int(tr[sub]["write"]),
                                                 "launches_counted": tr[sub]["launches"], "calibration": tr["calibration"],
                                                 "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate "
                                                           "passes over the same train step, KB -> bytes, scaled by the factors "
                                                           "measured on fi_calib_copy_kernel in the same run",
                                                 "collection_s": round(time.time() - t_p, 1)}
                if roof_roi is not None and roof_roi.get("traffic"):
                    roof_roi["traffic_GBps"] = round(roof_roi["traffic"] / (roof_roi["avg_launch_us"] * 1e-6) / 1e9, 1)
                    roof_roi["traffic_frac_hbm"] = round(roof_roi["traffic_GBps"] / HBM_PEAK_GBPS, 4)
        for v in kern.values():
            v.pop("_key", None)
        if roof_roi is not None:
            # lead with the memory-side figure (PMC traffic / duration) when there is one; B_min and the de-duplicated
            # bytes stay next to it.  Then the north star's own shape (every per-kernel sum of the step has been read).
            if roof_roi.get("traffic_frac_hbm") is not None:
                roof_roi = dict([("frac_hbm_memory_side", roof_roi["traffic_frac_hbm"]),
                                 ("frac_by_B_min", roof_roi["frac"])] + list(roof_roi.items()))
            if world == 1:
                try:
                    ns = roof_roi["north_star_shape"] = north_star_roialign(dev)
                    if not args.no_pmc:
                        # the memory side of the north star's own launches: FETCH_SIZE / WRITE_SIZE of the two kernels in a
                        # child that runs only them (calibrated on the copy kernel in the same process, as above)
                        names = {"channels_last": "crop_fwd_cl_kernel<7, 7", "nchw": "crop_fwd_flat_kernel<7, 7"}
                        tr2 = pmc_traffic(list(names.values()), [], timeout_s=120, child_flag="--pmc-child-roi")
                        for lay, sub in names.items():
                            if lay in ns and sub in tr2:
                                t_b = tr2[sub]["fetch"] + tr2[sub]["write"]
                                gbps = t_b / (ns[lay]["avg_launch_us"] * 1e-6) / 1e9
                                ns[lay].update(traffic=int(t_b), fetch_bytes=int(tr2[sub]["fetch"]), write_bytes=int(tr2[sub]["write"]),
                                               launches_counted=tr2[sub]["launches"], traffic_GBps=round(gbps, 1),
                                               frac_hbm_memory_side=round(gbps / HBM_PEAK_GBPS, 4),
                                               frac_by_B_min=ns[lay]["frac"])
                        if "error" in tr2:
                            ns["traffic_error"] = tr2["error"]
                        elif "calibration" in tr2:
                            ns["calibration"] = tr2["calibration"]
                except Exception as ex:
                    roof_roi["north_star_shape"] = {"error": repr(ex)}
        out = {
            "metric": "images/sec (train step, ResNet-101-FPN 1024^2, 512 RoIs)", "value": round(value, 4),
            "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.conv_precision == "fp32" else "%s (conv operands; fp32 accumulation, fp32 elsewhere)" % args.conv_precision,
            "data": "synthetic (seeded N(0,1)*64 images, 20 GT boxes/img, random-init weights, "
                    "GT-jittered proposals planted among RPN candidates before NMS -- 16 sets drawn before the timed "
                    "region and handed out in turn, resident in HBM like the images)",
            "config": {"workload": ("BASELINE configs[2]" if args.config == "cfg3" else
                                    "single-GPU slice of BASELINE configs[4] (1333x800 padded to a /64 multiple, SURVEY Q8)") +
                                   ": %s-FPN, %dx%d, %d images/GPU, %d RoIs/image, OT intertwiner "
                                   "on (Sinkhorn L=%d, 256 samples, 80 classes), full train step fwd+loss+bwd+clip+SGD"
                                   % (args.backbone, args.image_size, args.image_size, args.batch_per_gpu, args.rois,
                                      args.ot_L),
                       "global_batch": global_batch, "parallelism": "dp%d" % world,
                       "variant": ("mask head on positive slots only (dead-work elimination, not the reference's "
                                   "schedule)" if args.mask_head_on_positive_slots else
                                   ("reference schedule, dense backward (round-2 form)" if args.dense_backward else
                                    "reference schedule: every forward result of the reference is computed (all anchors' RPN "
                                    "outputs, every RoI's masks of every class); the backward pass computes the same "
                                    "gradients and skips the parts that are identically zero -- the RPN losses read 256 "
                                    "sampled anchors per image, the mask loss the positive RoIs' target-class masks "
                                    "(DESIGN.md section 3; --dense-backward runs the dense form)")),
                       "conv_stack": ("hand-written fp32 MFMA implicit-GEMM kernels (csrc/conv_igemm.hip)" if
                                      args.conv_precision == "fp32" else "hand-written 16-bit-operand (" + args.conv_precision + ") / fp32-accumulate MFMA "
                                      "kernels (csrc/conv_bf16.hip; layers with Cin % 32 != 0 on the fp32 kernels)") +
                                     "; full-window convs and nn.Linear on the library GEMM"},
            "losses": {k: round(float(v), 5) for k, v in terms.items()},
            "dense_backward_reference": dense_ref,
            "backward_check": backward_check,
            "configs4_slice": cfg4_slice,
            "issue": issue, "two_images_per_gpu_graph_ab": graph_ab,
            "roofline": roof, "roofline_roialign": roof_roi, "conv_stack": conv_stack, "nms": nms_obj, "sinkhorn": sk_obj,
            "timing": {"timed_region": "%d steps, no event recording / logging" % args.steps,
                       "profiled_pass": "%d further steps with HIP-event timing of every library kernel, weight gradients "
                                        "and the mask head's unread batch on the main stream so that every duration is "
                                        "the kernel alone on the chip (the timed region runs them on a second stream): "
                                        "%.2f ms/step" % (prof_steps, prof_elapsed / prof_steps * 1e3)},
            "kernels": kern,
        }
        if world > 1 or force_dp:
            out["data_parallel"] = {
                "engine": "one process per GPU; gradients live in one arena per rank and every ~64 MB bucket is a "
                          "contiguous slice of it, all-reduced IN PLACE on a side HIP stream from autograd hooks; one "
                          "~1 MB all-reduce of the intertwiner class statistics in forward",
                "buckets": len(sync.buckets), "bucket_bytes": sync.bucket_bytes(),
                "ms_per_step_per_rank": per_rank_ms, "rccl_ranks": rccl_ranks, "overlap": overlap,
                "streams": sync.stream_choice, "late_gradients": sync.late_gradients,
                "configs3_shape": configs3}
        if world == 1 and not args.no_cpu_baseline:
            try:
                entries = [e for e in log if e["pyramid"] and e["crop"] in (7, 14) and e["boxes"].size(0) ==
                           args.batch_per_gpu * args.rois][-2:]
                out["cpu_baseline"] = cpu_baseline(model, batch, entries, shape_log)
            except Exception as ex:   # the baseline must never take the headline down
                out["cpu_baseline"] = {"error": repr(ex)}
            if args.config == "cfg3" and args.conv_precision == "fp32" and not args.dense_backward:
                try:
                    out["cpu_baseline_configs0"] = cpu_baseline_configs0(dev)
                except Exception as ex:
                    out["cpu_baseline_configs0"] = {"error": repr(ex)}
        result_line = json.dumps(out)
    if world > 1 or force_dp:
        dist.destroy_process_group()
    return result_line


if __name__ == "__main__":
    main()
