"""Build the HIP extension (libcc_hip.so) in-tree with hipcc for gfx950. No JIT cache, no CPU variant."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcc_hip.so")
SOURCES = ["cc_engine.hip", "cc_eval.hip", "cc_kitti.hip", "cc_gt_labels.hip"]
DEPS = ["cc_engine.hip", "cc_eval.hip", "cc_kitti.hip", "cc_gt_labels.hip", os.path.join("..", "..", "include", "cc_kitti.h"), "cc_assoc_shared.h", "cc_assoc3.h", "cc_assocb.h", "cc_kernels.h", "cc_k_base.h", "cc_k_segcells.h", "cc_k_insert.h", "cc_k_segment.h", "cc_k_assoc_global.h", "cc_k_scan.h", "cc_k_assoc_lds.h", "cc_k_publish.h", "cc_device.h", "cc_math.h", os.path.join("..", "..", "include", "cc_hip.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall",
               "-Wno-unused-variable", "-Wno-bitwise-instead-of-logical", "-ldl"]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X extension cannot be built")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS if os.path.exists(os.path.join(CSRC, d)))


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    cmd = [hipcc(), *HIPCC_FLAGS, "-o", LIB, *[os.path.join(CSRC, s) for s in SOURCES]]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
