"""Which HIP streams share a hardware queue?  (GPU_MAX_HW_QUEUES, default 4, bounds the queues a process gets; streams
beyond that share one, and kernels of two streams on one queue run one after the other.)  A spin kernel on stream A, a tiny
kernel on stream B right behind it: B finishing before A's spin ends = different queues.
    python scripts/stream_queues.py [--pg]      (--pg: a 1-rank RCCL process group and one collective first)"""
import os
import sys

import torch

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
x = torch.zeros(1024, device=dev)
if "--pg" in sys.argv:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    dist.init_process_group("nccl", rank=0, world_size=1)
    dist.all_reduce(x)
    torch.cuda.synchronize()

# calibrate the spin
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record(); torch.cuda._sleep(1000000); t1.record(); torch.cuda.synchronize()
per_ms = 1000000 / t0.elapsed_time(t1)
SPIN = int(per_ms * 3)


def concurrent(a, b):
    """True when a kernel on b overtakes a 3 ms spin on a."""
    torch.cuda.synchronize()
    ea, eb, e0 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(a):
        e0.record(a)
        torch.cuda._sleep(SPIN)
        ea.record(a)
    with torch.cuda.stream(b):
        y = x + 1          # noqa: F841
        eb.record(b)
    torch.cuda.synchronize()
    return e0.elapsed_time(eb) < 0.5 * e0.elapsed_time(ea)


main = torch.cuda.current_stream()
pool = [torch.cuda.Stream() for _ in range(10)]
for st in pool:                 # a stream is bound to its hardware queue (created then, which takes milliseconds) at first use
    with torch.cuda.stream(st):
        x.add_(0.0)
torch.cuda.synchronize()
names = ["main"] + ["s%d" % i for i in range(len(pool))]
allst = [main] + pool
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES", "(default)"), " spin cycles per ms %.0f" % per_ms)
print("      " + " ".join("%4s" % n for n in names))
for i, a in enumerate(allst):
    row = []
    for j, b in enumerate(allst):
        row.append("   ." if i == j else ("   c" if concurrent(a, b) else "   S"))
    print("%5s " % names[i] + " ".join(row), flush=True)
if "--pg" in sys.argv:
    # the process group's own stream: does a collective overtake a spin on main / on s_i ?
    import torch.distributed as dist
    for n, a in zip(names, allst):
        torch.cuda.synchronize()
        side = pool[-1] if a is not pool[-1] else pool[-2]
        e0, ea, ew = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        with torch.cuda.stream(a):
            e0.record(a)
            torch.cuda._sleep(SPIN)
            ea.record(a)
        with torch.cuda.stream(side):
            w = dist.all_reduce(x, async_op=True)
            w.wait()
            ew.record(side)
        torch.cuda.synchronize()
        print("collective vs spin on %-5s: %s" % (n, "concurrent" if e0.elapsed_time(ew) < 0.5 * e0.elapsed_time(ea) else "SERIAL"))
if "--pg" in sys.argv:
    # The other direction: is work on stream X held up by a collective that is itself waiting (here: for a spin kernel on
    # the stream it was issued from)?  X = the default (null) stream and a non-null stream with a queue of its own.
    import torch.distributed as dist
    issue, other = pool[0], pool[1]          # s0, s1: different hardware queues (matrix above)
    for label, xs in (("default (null) stream", main), ("non-null stream s1", other)):
        torch.cuda.synchronize()
        e0, e_x, e_spin = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        with torch.cuda.stream(issue):
            e0.record(issue)
            torch.cuda._sleep(SPIN)
            e_spin.record(issue)
            w = dist.all_reduce(x, async_op=True)        # the process group's stream now waits for the spin on `issue`
        with torch.cuda.stream(xs):
            y = torch.ones(16, device=dev) + 1           # noqa: F841  issued AFTER the collective, independent of it
            e_x.record(xs)
        w.wait()
        torch.cuda.synchronize()
        held = e0.elapsed_time(e_x) > 0.5 * e0.elapsed_time(e_spin)
        print("a kernel on the %-22s issued behind a pending collective: %s" % (label, "HELD UP until the collective ran" if held else "runs at once"))
