"""Throughput of the KITTI frame -> firings kernels (cc_kitti_convert_frames): F frames per call, R calls.
Prints wall-clock per call (H2D of the .bin payloads from pageable memory included); run under
`rocprofv3 --kernel-trace --stats` for the per-kernel times (profiles/r01_kitti_kernel_stats.csv)."""
import sys
import time

import numpy as np
import torch

from continuous_clustering_amd import kitti

F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
stamps = np.array([0, 10**8, 2 * 10**8], dtype=np.uint64) + np.uint64(10**18)
rows, _ = kitti.synthetic_poses(3)
poses = np.stack([kitti.pose_from_line(r, kitti.CALIB_TR) for r in rows])
start, end = kitti.start_end_stamps(stamps)
bins = kitti.bin_transforms(stamps, poses, start[1], end[1], poses[1])
base = [kitti.synthetic_frame(s)[0] for s in range(4)]
d_xyz = torch.empty((F, 2200, 64, 3), dtype=torch.float32, device="cuda")
d_int = torch.empty((F, 2200, 64), dtype=torch.uint8, device="cuda")
d_org = torch.empty((F, 2200, 64), dtype=torch.int32, device="cuda")
frames = [dict(points=base[f % 4], stages=kitti.ALL_STAGES, start=start[1], end=end[1], bins=bins, d_xyz=d_xyz[f].data_ptr(),
               d_intensity=d_int[f].data_ptr(), d_original_index=d_org[f].data_ptr()) for f in range(F)]
conv = kitti.KittiConverter(max_frames=F, max_points=max(b.shape[0] for b in base))
conv.convert(frames)
conv.sync()
npts = sum(fr["points"].shape[0] for fr in frames)
for r in range(R):
    t = time.perf_counter()
    conv.convert(frames)
    conv.sync()
    dt = time.perf_counter() - t
    print(f"call {r}: {F} frames, {npts} points, {dt * 1e3:.2f} ms -> {F / dt:.0f} frames/s, {F * 2200 * 64 / dt / 1e6:.0f} Mcells/s (wall, H2D included)")
