"""CPU, build container only: pin constants and the inline helpers of the hot path against the REFERENCE'S OWN HEADERS, compiled as they lie.

general.hpp, point_types.hpp and utils/thread_pool.hpp need no Eigen, so — unlike cc.cpp (tests/test_reference_build.py, which skips here) —
they compile in this image. tests/csrc/reference_headers_probe.cpp static_asserts every CC_GP_* / CC_DBG_* value (include/cc_hip.h:45-62)
and the drop-in class's GP_* enum against the colour enum (general.hpp:208-357), RawPoint / RawPoints size and field offsets
(point_types.hpp:10-28) against the drop-in class's mirror, checks Point3D::operator- + lengthSquared (the association's distance test,
cc.cpp:638-641) against the oracle's close_enough and Point2D::length / lengthXY against the oracle's len2 on 2 M random pairs bit for bit,
and that ThreadPool::enqueue runs a job inline with num_threads == 0 (thread_pool.hpp:31-35,58-64: the single-threaded mode the oracle
restates). The names cc.hpp:15-22 gives the ground labels are checked as text. The GPU box has no /root/reference: there this test SKIPS.
It does not lift "parity unpinned"  — the algorithm itself needs Eigen3 — it takes the constants out of the unpinned set."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("CC_REFERENCE_ROOT", "/root/reference")
needs_reference = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "include", "continuous_clustering", "clustering", "general.hpp")),
                                     reason="the reference tree is not present on this machine")


@needs_reference
def test_constants_layouts_and_inline_helpers_match_the_reference_headers(tmp_path):
    exe = str(tmp_path / "reference_headers_probe")
    src = os.path.join(ROOT, "tests", "csrc", "reference_headers_probe.cpp")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", "-Wno-subobject-linkage", "-I" + os.path.join(REF, "include"), "-o", exe, src],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]  # (a static_assert that fails is a compile error naming the constant)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-2000:]
    assert re.search(r"close (\d+) ", r.stdout) and int(re.search(r"close (\d+) ", r.stdout).group(1)) > 100000, r.stdout  # both outcomes exercised
    assert " bad 0" in r.stdout


@needs_reference
def test_ground_label_names_of_the_reference_class_header():
    """cc.hpp:15-22 cannot be compiled here (it includes Eigen); the five enumerators are one line each: GP_X = COLOUR."""
    text = open(os.path.join(REF, "include", "continuous_clustering", "clustering", "continuous_clustering.hpp")).read()
    got = dict(re.findall(r"\b(GP_[A-Z_]+)\s*=\s*([A-Z]+)\s*,", text))
    assert got == {"GP_UNKNOWN": "WHITE", "GP_GROUND": "GREEN", "GP_OBSTACLE": "RED", "GP_EGO_VEHICLE": "MAGENTA", "GP_FOG": "LIGHTGRAY"}
    ours = open(os.path.join(ROOT, "include", "cc_hip.h")).read()
    for name, colour in got.items():
        assert re.search(r"CC_%s = \d+,\s*/\* %s \*/" % (name, colour), ours), (name, colour)
