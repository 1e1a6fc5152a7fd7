#!/usr/bin/env python3
"""Regenerate the golden fixtures tests/golden/<case>.npz.

PROVENANCE: the reference (UniBwTAS/continuous_clustering) ships no golden vectors and its core cannot be
compiled in this image (it needs Eigen3, absent), so these fixtures are produced by the CPU oracle
(oracle/cc_oracle.cpp), a restatement of the reference's single-threaded algorithm. They are regression vectors that
pin the oracle's behaviour ("parity unpinned" with respect to the reference itself) and let the GPU
tests check the HIP path against committed data. Each file holds the inputs (firings, poses, config) and the
expected outputs (event log, published columns).

    python tests/golden/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cases  # noqa: E402
import util  # noqa: E402
from continuous_clustering_amd import capi  # noqa: E402

OUT_FIELDS = ["ground_point_label", "debug_ground_point_label", "is_ignored", "id", "tree_root_global_column", "tree_root_row",
              "source_firing", "distance", "inclination_angle"]
EXTRA_FIELDS = {"g_s64_translate": ["continuous_azimuth_angle", "x", "y", "z"]}


def main():
    for name in cases.GOLDEN_CASES:
        stream, cfg, tf = cases.build_case(name)
        o, rc = util.run_oracle(stream, cfg, tf)
        assert rc == 0
        ev = o.drain_events()
        frm, to = o.published_range()
        cols = o.read_published(frm, to, OUT_FIELDS + EXTRA_FIELDS.get(name, []))
        cols["id"] = cols["id"].astype(np.uint32)
        cols["source_firing"] = cols["source_firing"].astype(np.int32)
        cols["tree_root_global_column"] = cols["tree_root_global_column"].astype(np.int32)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(
            path, xyz=stream.xyz, intensity=stream.intensity, poses=stream.poses,
            config=np.frombuffer(bytes(cfg), dtype=np.uint8), num_rows=np.int32(stream.sensor.num_rows),
            robot_tf=np.zeros(0) if tf is None else tf, events=ev, first_column=np.int64(frm),
            state=np.array([o.state()[k] for k in util.STATE_FIELDS], dtype=np.int64), **{"out_" + k: v for k, v in cols.items()})
        print(name, os.path.getsize(path) // 1024, "KiB", len(ev), "events", to - frm + 1, "columns")


if __name__ == "__main__":
    main()
