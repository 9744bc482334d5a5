"""Build libfi_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc.

In-tree build: the shared object is written next to this file so that it travels
to the GPU box with the repository snapshot.  Usage:

    python -m feature_intertwiner_amd.build [--force]
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libfi_hip.so")
SOURCES = ["fi_core.hip", "crop_and_resize.hip", "roi_pool.hip", "nms.hip", "sinkhorn.hip",
           "class_mean.hip", "conv_igemm.hip", "conv1x1_ring.hip", "conv_bf16.hip", "conv_f16.hip", "sgd.hip", "glue.hip", "proposal.hip", "targets.hip", "losses.hip", "dev_stage.hip", "meta_stats.hip"]
HEADERS = ["fi_common.h", os.path.join("..", "..", "include", "fi_capi.h")]

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    # bit-exact coordinate / IoU arithmetic: no fused multiply-add contraction
    "-ffp-contract=off",
    # hardware fp32 atomic add (global_atomic_add_f32) for the scatter kernels
    "-munsafe-fp-atomics",
    "-Wall", "-Wno-unused-function",
    # the ring kernel's LDS-DMA asm names m0 as clobbered (a reserved register: clang warns per use)
    "-Wno-inline-asm",
] + os.environ.get("FI_EXTRA_HIPCC_FLAGS", "").split()          # e.g. -DFI_PROBE_1X1 (scripts/c4_probe.sh)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    objs = []
    for s in srcs:
        o = s[:-4] + ".o"
        deps = [s] + hdrs + ([os.path.join(CSRC, "conv_bf16.hip")] if s.endswith("conv_f16.hip") else [])
        if force or _stale(o, deps):
            cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(LIB_PATH, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
