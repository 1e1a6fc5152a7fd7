"""CPU: a second, independently written restatement of continuous range-image insertion (cc.cpp:105-292) and ground-point
segmentation (cc.cpp:294-624), used to cross-check oracle/cc_oracle.cpp (which has one author and one reading).

Written from the algorithm description in SURVEY.md 8a / Appendix A, on purpose with a different structure than the oracle:
 * the range image is a plain dict  global column -> per-row numpy arrays  (no ring buffer, no clearing, no local columns);
 * a firing is processed with numpy vector operations over its rows where the reference loops (the collision rule is per row and
   rows never interact inside one firing, so this is the same computation);
 * segmentation is a per-column function with an explicit small state machine over labels, fed by the insertion's output.
float32 / float64 evaluation order follows the reference (Eigen's unrolled 3x4 products, std::atan2 / std::asin on floats through
glibc via ctypes — numpy's own float32 arctan2 may take a SIMD path with different last bits).

What is compared with the oracle on several parity cases: every published column's geometry bit patterns, global column index, source
firing, continuous azimuth (also of empty cells), supplemented inclination, ground / debug labels and ignore flags, plus the ring
bookkeeping scalars. The association / finished-cluster part has its own independent check in test_oracle_properties.py."""
import ctypes
import ctypes.util
import math

import numpy as np
import pytest

import cases
import util
from continuous_clustering_amd import capi

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.atan2f.restype = ctypes.c_float
_libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]
_libm.asinf.restype = ctypes.c_float
_libm.asinf.argtypes = [ctypes.c_float]
F32 = np.float32


def atan2f(y, x):
    return F32(_libm.atan2f(float(y), float(x)))


def asinf(v):
    return F32(_libm.asinf(float(v)))


def cvttss2si(v):
    """static_cast<int>(float) on x86-64."""
    if not (v == v) or v >= 2147483648.0 or v < -2147483648.0:
        return -2147483648
    return int(v)  # truncation toward zero


WHITE, GRAY, ORANGE, GREEN, YELLOWGREEN, YELLOW, RED, DARKRED, VIOLET, LIGHTGRAY = range(10)


class Independent:
    """Insertion + ground segmentation of one stream; columns are kept by GLOBAL index."""

    def __init__(self, cfg, rows, robot_tf):
        self.cfg, self.R, self.NC = cfg, rows, cfg.num_columns
        self.width = F32(F32(2 * math.pi) / F32(self.NC))
        self.cells = {}             # gcol -> dict of per-row arrays
        self.prev_rear, self.prev_fore, self.first_unfinished = 0, -1, -1
        self.ring_start, self.ring_end = -1, -1
        self.reset_required = False
        self.table = np.full(rows, np.nan, dtype=F32)  # inclination steps between neighbouring lasers, carried across columns
        self.robot = np.asarray(robot_tf, dtype=np.float64).reshape(3, 4)
        self.segmented = {}         # gcol -> outputs
        self.n_firings = 0
        # label value tables (include/cc_hip.h mirrors the reference's enums)
        self.DBG = self._debug_values()

    @staticmethod
    def _debug_values():
        import re, os
        txt = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "cc_hip.h")).read()
        out = {}
        for name in ("WHITE", "GRAY", "ORANGE", "GREEN", "YELLOWGREEN", "YELLOW", "RED", "DARKRED", "VIOLET", "LIGHTGRAY"):
            out[name] = int(re.search(r"CC_DBG_%s\s*=\s*(\d+)" % name, txt).group(1))
        return out

    def column(self, g):
        c = self.cells.get(g)
        if c is None:
            R = self.R
            c = dict(x=np.full(R, np.nan, F32), y=np.full(R, np.nan, F32), z=np.full(R, np.nan, F32), dist=np.full(R, np.nan, F32),
                     incl=np.full(R, np.nan, F32), caz=np.full(R, np.nan, np.float64), src=np.full(R, -1, np.int64), inten=np.zeros(R, np.uint8),
                     filled=np.zeros(R, bool))
            self.cells[g] = c
        return c

    # ---- cc.cpp:105-292 ---------------------------------------------------------------------------------------------
    def add_firing(self, xyz, inten, pose12):
        T = np.asarray(pose12, dtype=np.float64).reshape(3, 4)
        R, NC = self.R, self.NC
        seq = self.n_firings
        self.n_firings += 1
        p = xyz.astype(np.float64)
        valid = ~np.isnan(xyz[:, 0])
        # odom = R p + t, evaluated ((r0 x + r1 y) + r2 z) + t like Eigen's unrolled product
        odom = np.stack([((T[i, 0] * p[:, 0] + T[i, 1] * p[:, 1]) + T[i, 2] * p[:, 2]) + T[i, 3] for i in range(3)], axis=1)
        rel = odom - T[:, 3][None, :]
        dist = np.sqrt((rel[:, 0] * rel[:, 0] + rel[:, 1] * rel[:, 1]) + rel[:, 2] * rel[:, 2]).astype(F32)
        prev_rot = self.prev_rear // NC
        prev_cir = self.prev_rear % NC
        half = NC // 2
        rear, fore = -1, -1
        for r in np.nonzero(valid)[0]:      # (per row: the rows of one firing never touch the same cell)
            az = atan2f(xyz[r, 1], xyz[r, 0])
            inc_az = F32(F32(-az) + F32(math.pi)) if self.cfg.sensor_is_clockwise else F32(az + F32(math.pi))
            cir = cvttss2si(F32(inc_az / self.width))
            g = prev_rot * NC + cir
            off = 0
            diff = cir - prev_cir
            if diff < -half:
                g += NC
                off = 1
            elif self.prev_rear > 0 and diff > half:
                g -= NC
                off = -1
            if g < 0:
                continue  # (undefined behaviour in the reference; both restatements drop the return)
            caz = (2 * math.pi) * float(prev_rot + off) + float(inc_az)
            d = dist[r]
            cell = self.column(g)
            if cell["filled"][r] and not np.isnan(d):
                nxt = self.column(g + 1)
                if not nxt["filled"][r]:
                    cell, g = nxt, g + 1
            if cell["filled"][r] and (np.isnan(d) or d >= cell["dist"][r]):
                continue
            if not (self.first_unfinished >= 0 and g < self.first_unfinished):
                cell["x"][r], cell["y"][r], cell["z"][r] = F32(odom[r, 0]), F32(odom[r, 1]), F32(odom[r, 2])
                cell["dist"][r] = d
                cell["incl"][r] = asinf(F32(F32(rel[r, 2]) / d))
                cell["caz"][r] = caz
                cell["src"][r] = seq
                cell["inten"][r] = inten[r]
                cell["filled"][r] = not np.isnan(d)
            rear = g if rear < 0 or g < rear else rear
            fore = g if fore < 0 or g > fore else fore
        if rear >= 0 and fore >= 0:
            if fore - rear > NC // 2:
                self.reset_required = True
                return
            self.prev_rear = max(self.prev_rear, rear)
            self.prev_fore = max(self.prev_fore, fore)
        if self.prev_fore < 0:
            return
        if self.ring_start == -1:
            self.ring_start = self.prev_rear
        self.ring_end = max(self.ring_end, self.prev_fore)
        if self.first_unfinished == -1:
            self.first_unfinished = self.prev_rear
        while self.first_unfinished < self.prev_rear:
            self.segment(self.first_unfinished, T)
            self.first_unfinished += 1

    # ---- cc.cpp:294-624 ---------------------------------------------------------------------------------------------
    def segment(self, g, T):
        cfg, R = self.cfg, self.R
        c = self.column(g)
        # robot_from_odom = robot_from_sensor * inverse(odom_from_sensor); the inverse of a rigid transform is [R^T | -R^T t]
        Rt = T[:, :3].T.copy()
        it = np.array([((-Rt[i, 0]) * T[0, 3] + (-Rt[i, 1]) * T[1, 3]) + (-Rt[i, 2]) * T[2, 3] for i in range(3)])
        A = self.robot
        er = np.array([[(A[i, 0] * Rt[0, j] + A[i, 1] * Rt[1, j]) + A[i, 2] * Rt[2, j] for j in range(3)] for i in range(3)])
        et = np.array([((A[i, 0] * it[0] + A[i, 1] * it[1]) + A[i, 2] * it[2]) + A[i, 3] for i in range(3)])
        sensor = T[:, 3].astype(F32)
        h_sensor_ground = F32(F32(-F32(A[2, 3])) + F32(cfg.height_ref_to_ground_))
        incl = c["incl"].copy()
        caz = c["caz"].copy()
        ground = np.full(R, capi.GP_UNKNOWN, np.uint8)
        debug = np.full(R, self.DBG["WHITE"], np.uint8)
        state = dict(first_obstacle=False, first_found=False, lg=(F32(0), h_sensor_ground), prev=None, prev_label=None)
        prev_incl = F32(0)
        plane = {}  # row -> (x in the azimuth plane, z) relative to the sensor
        D = self.DBG
        for r in range(R - 1, -1, -1):
            raw = c["incl"][r]
            step = F32(raw - prev_incl)
            if not np.isnan(step):
                self.table[r] = step
            prev_incl = raw
            if np.isnan(c["dist"][r]):
                if cfg.supplement_inclination_angle_for_nan_cells and r < R - 1:
                    incl[r] = F32(incl[r + 1] + self.table[r])
                caz[r] = (float(g) + 0.5) * float(self.width)
                continue
            if (cfg.fog_filtering_enabled and int(c["inten"][r]) < (cfg.fog_filtering_intensity_below & 0xff) and
                    c["dist"][r] < F32(cfg.fog_filtering_distance_below) and incl[r] > F32(cfg.fog_filtering_inclination_above)):
                ground[r], debug[r] = capi.GP_FOG, D["LIGHTGRAY"]
                continue
            q = np.array([c["x"][r], c["y"][r], c["z"][r]], dtype=np.float64)
            e = [((er[i, 0] * q[0] + er[i, 1] * q[1]) + er[i, 2] * q[2]) + et[i] for i in range(3)]
            if (e[0] < cfg.length_ref_to_front_end_ and e[0] > cfg.length_ref_to_rear_end_ and e[1] < cfg.width_ref_to_left_mirror_ and
                    e[1] > cfg.width_ref_to_right_mirror_ and e[2] < cfg.height_ref_to_maximum_ and e[2] > cfg.height_ref_to_ground_):
                ground[r], debug[r] = capi.GP_EGO_VEHICLE, D["VIOLET"]
                continue
            ux, uy, uz = F32(c["x"][r] - sensor[0]), F32(c["y"][r] - sensor[1]), F32(c["z"][r] - sensor[2])
            cur = (F32(np.sqrt(F32(F32(ux * ux) + F32(uy * uy)))), uz)
            plane[r] = cur
            if not state["first_found"]:
                state["first_found"] = True
                h = F32(cur[1] - h_sensor_ground)
                if h > F32(cfg.first_ring_as_ground_min_allowed_z_diff) and h < F32(cfg.first_ring_as_ground_max_allowed_z_diff):
                    ground[r], debug[r] = capi.GP_GROUND, D["GRAY"]
                    state["lg"] = cur
                    state["first_obstacle"] = False
                else:
                    ground[r], debug[r] = capi.GP_OBSTACLE, D["ORANGE"]
                    state["first_obstacle"] = True
                state["prev"], state["prev_label"] = cur, debug[r]
                continue
            pv, lg = state["prev"], state["lg"]
            with np.errstate(divide="ignore", invalid="ignore"):
                p2c = (F32(cur[0] - pv[0]), F32(cur[1] - pv[1]))
                slope_prev = F32(p2c[1] / p2c[0])
                l2c = (F32(cur[0] - lg[0]), F32(cur[1] - lg[1]))
                slope_lg = F32(l2c[1] / l2c[0])
            flat_prev = abs(slope_prev) < F32(cfg.max_slope) and p2c[0] > 0
            flat_prev = flat_prev and (not cfg.use_terrain or p2c[0] < 5)
            flat_lg = abs(slope_lg) < F32(cfg.max_slope) and l2c[0] > 0
            if not state["first_obstacle"] and flat_prev:
                ground[r], debug[r] = capi.GP_GROUND, D["GREEN"]
            elif not cfg.use_terrain:
                if state["first_obstacle"] and flat_prev and flat_lg:
                    ground[r], debug[r] = capi.GP_GROUND, D["YELLOWGREEN"]
                elif (abs(l2c[0]) < F32(cfg.ground_because_close_to_last_certain_ground_max_dist_diff) and
                      abs(l2c[1]) < F32(cfg.ground_because_close_to_last_certain_ground_max_z_diff)):
                    ground[r], debug[r] = capi.GP_GROUND, D["YELLOW"]
            if ground[r] != capi.GP_GROUND:
                ground[r], debug[r] = capi.GP_OBSTACLE, D["RED"]
                below = r + 1
                while below < R:
                    # (rows that were skipped keep NaN coordinates: their difference compares false, like in the reference)
                    bx = plane[below][0] if below in plane else F32(np.nan)
                    if debug[below] == D["YELLOW"] or (ground[below] == capi.GP_GROUND and
                                                       abs(F32(cur[0] - bx)) < F32(cfg.obstacle_because_next_certain_obstacle_max_dist_diff)):
                        if ground[below] == capi.GP_GROUND:
                            ground[below], debug[below] = capi.GP_OBSTACLE, D["DARKRED"]
                        below += 1
                    else:
                        break
            state["first_obstacle"] = state["first_obstacle"] or ground[r] == capi.GP_OBSTACLE
            if debug[r] in (D["GREEN"], D["YELLOWGREEN"]):
                if (slope_prev > F32(cfg.last_ground_point_slope_higher_than) and
                        abs(p2c[0]) < F32(cfg.last_ground_point_distance_smaller_than) and state["prev_label"] != D["YELLOW"]):
                    state["lg"] = cur
            state["prev"], state["prev_label"] = cur, debug[r]
        ignored = np.zeros(R, np.uint8)
        for r in range(R):
            if np.isnan(c["dist"][r]) or ground[r] != capi.GP_OBSTACLE:
                ignored[r] = 1
            elif float(c["dist"][r]) < 1.0 * float(F32(cfg.max_distance)):
                ignored[r] = 1
            elif (cfg.ignore_points_with_too_big_inclination_angle_diff and r < R - 1 and
                  atan2f(F32(cfg.max_distance), c["dist"][r]) < self.table[r]):
                ignored[r] = 1
            elif cfg.ignore_points_in_chessboard_pattern and ((g % 2 == 0) != (r % 2 == 0)):
                ignored[r] = 1
        self.segmented[g] = dict(ground=ground, debug=debug, ignored=ignored, incl=incl, caz=caz)


INDEPENDENT_CASES = ["g_s64_translate", "g_s64_fog_and_ego", "g_s128_offsets", "s64_turn", "s64_counterclockwise", "s64_dropouts",
                     "s64_no_supplement_no_incl_ignore", "s32_small_sensor", "j_s64_jitter", "j_s64_jitter_wide", "j_s128_offsets_jitter", "s64_deep_lookback"]


@pytest.mark.parametrize("name", INDEPENDENT_CASES)
def test_second_restatement_of_insertion_and_segmentation_agrees_with_the_oracle(name, oracle_lib):
    stream, cfg, tf = cases.build_case(name)
    n = min(stream.n_firings, 1100)
    from oracle.pyoracle import Oracle, IDENTITY_TF
    robot = IDENTITY_TF if tf is None else tf
    o = Oracle(cfg, stream.sensor.num_rows, robot)
    assert o.add_firings(stream.xyz[:n], stream.intensity[:n], stream.poses[:n]) == 0
    ind = Independent(cfg, stream.sensor.num_rows, robot)
    for f in range(n):
        ind.add_firing(stream.xyz[f], stream.intensity[f], stream.poses[f])
    so = o.state()
    assert so["reset_required"] == int(ind.reset_required)
    assert so["first_unfinished_global_column_index"] == ind.first_unfinished
    assert so["ring_buffer_end_global_column_index"] == ind.ring_end
    lo, hi = o.published_range()
    assert hi - lo > stream.sensor.num_columns // 2
    ref = o.read_published(lo, hi)
    k = 0
    for g in range(lo, hi + 1):
        c, s = ind.cells[g] if g in ind.cells else ind.column(g), ind.segmented[g]
        row = g - lo
        has = ~np.isnan(c["dist"])
        for mine, theirs in ((c["x"], "x"), (c["y"], "y"), (c["z"], "z"), (c["dist"], "distance"), (s["incl"], "inclination_angle"),
                             (s["caz"], "continuous_azimuth_angle")):
            util.assert_float_equal(f"{theirs} of column {g}", np.asarray(mine), ref[theirs][row])
        assert np.array_equal(np.where(has, c["src"], -1), ref["source_firing"][row]), g
        assert (ref["global_column_index"][row] == g).all()
        assert np.array_equal(s["ground"], ref["ground_point_label"][row]), (g, s["ground"], ref["ground_point_label"][row])
        assert np.array_equal(s["debug"], ref["debug_ground_point_label"][row]), g
        assert np.array_equal(s["ignored"], ref["is_ignored"][row]), g
        k += 1
    assert k == hi - lo + 1
