"""Build-time audit of csrc/conv1x1_ring.hip's inline asm (CPU: hipcc cross-compiles gfx950).  The kernel issues its weight
loads and LDS-DMA pieces as asm so that the waits in its K loop are the counted ones it writes; the compiler knows nothing
about when an asm load's destination registers are actually written.  The invariant that makes that safe: between an asm
`global_load_dwordx4 v[a:b], ...` and the next asm `s_waitcnt vmcnt(N)` NO compiler-generated instruction may read or write
v[a:b] (a register-allocator copy of a value that has not landed would move garbage).  Also: no scratch (spills) in the
kernels, and every asm VMEM instruction is preceded by its hazard nops."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "feature_intertwiner_amd", "csrc", "conv1x1_ring.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def _all_vregs(line):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", line):
        out |= _regs(tok)
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_ring_kernel_asm_loads_are_not_touched_before_their_wait():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "ring.s")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                               "-munsafe-fp-atomics", "-Wno-inline-asm", "-S", "--cuda-device-only", SRC, "-o", out])
        text = open(out).read()
    kernels = re.findall(r"^(_ZN\S*conv1x1_ring_kernel\S*):[^\n]*\n(.*?)\n\.Lfunc_end", text, re.S | re.M)
    assert len(kernels) == 4
    for name, body in kernels:
        pending = {}            # vreg -> line number of the asm load that targets it
        in_asm = False
        n_loads = n_waits = 0
        for ln, line in enumerate(body.split("\n")):
            t = line.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t.startswith(";") or t.startswith("."):
                continue
            if in_asm:
                if t.startswith("global_load_dwordx4"):
                    dst = t.split()[1].rstrip(",")
                    for r in _regs(dst):
                        pending[r] = ln
                    n_loads += 1
                elif t.startswith("s_waitcnt vmcnt"):
                    pending.clear()     # every counted wait in this kernel covers all outstanding asm register loads
                    n_waits += 1
                continue
            touched = _all_vregs(t) & set(pending)
            assert not touched, "%s: compiler instruction touches in-flight asm load registers %s: %s" % (name, sorted(touched), t)
        assert n_loads >= 8 and n_waits >= 4, (name, n_loads, n_waits)
    # no scratch in any instantiation
    for m in re.finditer(r"\.name:\s+(\S*conv1x1_ring_kernel\S*)(.*?)\.vgpr_spill_count:\s+(\d+)", text, re.S):
        assert int(m.group(3)) == 0, m.group(1)
    # every asm VMEM instruction is preceded by its hazard nops
    for blk in re.findall(r";;#ASMSTART\n(.*?);;#ASMEND", text, re.S):
        if "global_" in blk:
            assert blk.strip().startswith("s_nop 4"), blk


NMS_SRC = os.path.join(ROOT, "feature_intertwiner_amd", "csrc", "nms.hip")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_wide_nms_scan_asm_loads():
    """csrc/nms.hip, nms_scan_wide_kernel: every global load is inline asm with a hand-counted wait.  Audited on the ISA:
      * the compiler inserted no vmcnt wait of its own into the kernel (it sees no vector memory load there);
      * between an asm load and the counted wait that covers it no compiler instruction touches the destination
        registers (loads retire in order: `s_waitcnt vmcnt(N)` completes all but the youngest N);
      * the scalar base of an asm load is never written by v_readlane / v_readfirstlane within 8 instructions before it
        (VALU-written SGPR -> VMEM needs 5 wait states the compiler does not know to insert for asm);
      * no scratch."""
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "nms.s")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                               "-munsafe-fp-atomics", "-Wno-inline-asm", "-S", "--cuda-device-only",
                               "-I", os.path.join(ROOT, "include"), NMS_SRC, "-o", out])
        text = open(out).read()
    kernels = re.findall(r"^(_ZN\S*nms_scan_wide_kernel\S*):[^\n]*\n(.*?)\n\.Lfunc_end", text, re.S | re.M)
    assert len(kernels) == 2
    for name, body in kernels:
        pending = []            # asm loads in issue order: (set of vregs, line)
        in_asm = False
        n_loads = n_waits = 0
        recent = []             # the last compiler instructions (for the SGPR hazard)
        for ln, line in enumerate(body.split("\n")):
            t = line.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t.startswith(";"):
                continue
            if t.startswith("."):
                continue
            if in_asm:
                if t.startswith("global_load_dwordx"):
                    ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
                    pending.append((_regs(ops[0]), ln))
                    n_loads += 1
                    m = re.fullmatch(r"s\[(\d+):(\d+)\]", ops[2])
                    if m:
                        base = {int(m.group(1)), int(m.group(2))}
                        for prev in recent[-8:]:
                            w = re.match(r"v_(?:readfirstlane|readlane)_b32 s(\d+)", prev)
                            assert not (w and int(w.group(1)) in base), (name, prev, t)
                elif t.startswith("s_waitcnt vmcnt"):
                    n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
                    pending = pending[len(pending) - n:] if n else []
                    n_waits += 1
                continue
            assert "s_waitcnt vmcnt" not in t, "%s: a compiler-inserted vmcnt wait: %s" % (name, t)
            if t.startswith("s_endpgm") or t.startswith("s_branch") or t.startswith("s_setpc"):
                pending = []    # (linear listing: what follows is another path)
            busy = set().union(*[r for r, _ in pending]) if pending else set()
            touched = _all_vregs(t) & busy
            assert not touched, "%s: compiler instruction touches in-flight asm load registers %s: %s" % (name, sorted(touched), t)
            recent.append(t)
        assert n_loads >= 20 and n_waits >= 8, (name, n_loads, n_waits)
    for m in re.finditer(r"\.name:\s+(\S*nms_scan_wide_kernel\S*)(.*?)\.vgpr_spill_count:\s+(\d+)", text, re.S):
        assert int(m.group(3)) == 0, m.group(1)
