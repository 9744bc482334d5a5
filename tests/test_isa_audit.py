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
