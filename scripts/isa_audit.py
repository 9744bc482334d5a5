"""Static VALU : MFMA ratio of the main loop of every MFMA kernel in a HIP source file.

    python scripts/isa_audit.py feature_intertwiner_amd/csrc/conv_igemm.hip [name substring ...]

Compiles the file to gfx950 assembly (hipcc -S --cuda-device-only), takes for each kernel the innermost
`Depth=1` loop with the most v_mfma instructions and counts instruction classes in it.  On CDNA4 a wavefront's
v_mfma and the other vector instructions of its SIMD do not overlap, so `valu / mfma`
times 4 issue cycles is MFMA time lost per MFMA (a 32x32x2 fp32 MFMA occupies the pipe for 64 cycles, a
32x32x16 bf16 one for 32).  Counts are static: rare branches inside the loop are included."""
import os
import re
import subprocess
import sys
import tempfile

src = sys.argv[1]
keys = sys.argv[2:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics",
                       "-S", "--cuda-device-only", "-I" + os.path.join(root, "include"),
                       "-I" + os.path.join(root, "feature_intertwiner_amd", "csrc"), src, "-o", asm],
                      stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
funcs, cur = {}, None
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        cur = m.group(1)
        funcs[cur] = [i, None]
    if l.startswith(".Lfunc_end") and cur:
        funcs[cur][1] = i
        cur = None


def demangle(n):
    try:
        d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except OSError:
        d = n
    return re.sub(r"\(anonymous namespace\)::|^void ", "", d).split("(")[0]


print("%-70s %5s %5s %9s %4s %5s %5s" % ("kernel", "mfma", "valu", "valu/mfma", "lds", "vmem", "salu"))
for n, (a, b) in funcs.items():
    body = lines[a:b]
    if sum(1 for l in body if "v_mfma" in l) < 4:
        continue
    d = demangle(n)
    if keys and not any(k in d for k in keys):
        continue
    best = None
    for h in [i for i, l in enumerate(body) if "Loop Header" in l and "Depth=1" in l]:
        lab = body[h].split(":")[0]
        ends = [i for i, l in enumerate(body) if i > h and re.search(r"s_c?branch\w*\s+" + re.escape(lab) + r"\b", l)]
        if not ends:
            continue
        reg = body[h:max(ends) + 1]
        m = sum(1 for l in reg if "v_mfma" in l)
        if m and (best is None or m > best[0]):
            best = (m, reg)
    if not best:
        continue
    m, reg = best
    ins = [l.split()[0] for l in reg if l.startswith("\t") and not l.strip().startswith((";", "."))]
    valu = sum(1 for x in ins if x.startswith("v_") and "mfma" not in x)
    print("%-70s %5d %5d %9.2f %4d %5d %5d" % (d[:70], m, valu, valu / m, sum(1 for x in ins if x.startswith("ds_")),
                                              sum(1 for x in ins if x.startswith(("global_", "buffer_"))),
                                              sum(1 for x in ins if x.startswith("s_"))))
