#!/usr/bin/env python3
"""Per-kernel resource / occupancy table of libcc_hip.so's device code.

Compiles every .hip source for gfx950 (device side only) with -Rpass-analysis=kernel-resource-usage and prints, per kernel,
what the code object asks of a compute unit: VGPRs, AGPRs, SGPRs, scratch, LDS (static; dynamic LDS is chosen by the host at
launch and listed from cc_engine.hip's launch sites by hand in profiles/), the compiler's occupancy bound in wavefronts per
SIMD, and spills.  Needs no GPU.  Usage: python tools/kernel_resources.py [> profiles/r06_kernel_resources.txt]
"""
from __future__ import annotations

import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "continuous_clustering_amd", "csrc")
sys.path.insert(0, ROOT)
from continuous_clustering_amd import build as hip_build  # noqa: E402

KEYS = ["TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    short = []
    for d in out:
        d = re.sub(r"^void ", "", d)
        m = re.match(r"(?:cck::|cc\w*::)?([\w:]+)(<.*>)?\(", d)
        if m:
            nm = m.group(1).split("::")[-1]
            tp = m.group(2) or ""
            tp = re.sub(r"\(cc\w+::\w+\)", "", tp)
            short.append(nm + tp)
        else:
            short.append(d[:60])
    return short


def main() -> int:
    rows = []
    for src in hip_build.SOURCES:
        flags = [f for f in hip_build.HIPCC_FLAGS if f not in ("-shared", "-ldl")]
        cmd = [hip_build.hipcc(), *flags, "-c", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", os.path.join(CSRC, src)]
        err = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
        cur = None
        for line in err.splitlines():
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = {"name": m.group(1), "src": src}
                rows.append(cur)
                continue
            m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass", line)
            if m and cur is not None:
                cur[m.group(1).strip()] = m.group(2)
    names = demangle([r["name"] for r in rows])
    print("# per-kernel resources of libcc_hip.so's gfx950 code objects (hipcc -Rpass-analysis=kernel-resource-usage; tools/kernel_resources.py)")
    print("# flags:", " ".join(hip_build.HIPCC_FLAGS))
    print("# gfx950: 512 VGPRs per SIMD lane (unified VGPR+AGPR file, allocation granule 8), 8 wavefronts per SIMD at most, 160 KB LDS per CU, 4 SIMDs per CU")
    print("# waves/SIMD = the compiler's bound from registers alone (floor(512 / (VGPRs + AGPRs rounded up to 8)), <= 8); LDS = static bytes per block (dynamic LDS: see the launch table below)")
    hdr = f"{'kernel':<58} {'source':<16} {'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch':>8} {'LDS':>7} {'waves/SIMD':>10} {'sgpr spill':>10} {'vgpr spill':>10}"
    print(hdr)
    for r, n in sorted(zip(rows, names), key=lambda x: (x[0]["src"], x[1])):
        print(f"{n[:58]:<58} {r['src']:<16} {r.get('VGPRs','?'):>5} {r.get('AGPRs','?'):>5} {r.get('TotalSGPRs','?'):>5} {r.get('ScratchSize [bytes/lane]','?'):>8} "
              f"{r.get('LDS Size [bytes/block]','?'):>7} {r.get('Occupancy [waves/SIMD]','?'):>10} {r.get('SGPRs Spill','?'):>10} {r.get('VGPRs Spill','?'):>10}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
