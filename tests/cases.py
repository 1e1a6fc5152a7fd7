"""Named parity cases shared by the CPU (oracle / golden) and GPU (engine vs oracle) tests.

Every case is (stream, config, robot_from_sensor or None). Streams are small (seconds for the CPU oracle)."""
from __future__ import annotations

import numpy as np

from continuous_clustering_amd import capi, synth
from continuous_clustering_amd.synth import Motion, SceneModel, SensorModel


def _kitti(num_columns=2200, **over):
    c = capi.Config.kitti()
    c.num_columns = num_columns
    for k, v in over.items():
        setattr(c, k, v)
    return c


def _vls(num_columns=1700, **over):
    c = capi.Config.vls128()
    c.num_columns = num_columns
    for k, v in over.items():
        setattr(c, k, v)
    return c


def _s64(cols):
    return SensorModel(num_rows=64, num_columns=cols)


def _s128(cols, offsets=True):
    s = SensorModel.s128(offsets)
    s.num_columns = cols
    return s


def jitter_azimuth(stream, seed, amp_columns=0.8):
    """Rotate every firing about the sensor's z axis by a random fraction of a column: firings then repeat columns (cell collisions,
    the shift-to-the-next-column rule cc.cpp:188-202, refusals cc.cpp:204-206), skip columns (empty columns) and step backwards
    (returns behind the first unfinished column, cc.cpp:209-219) — what the evenly stepping synthetic sensor never does."""
    rng = np.random.default_rng(seed)
    w = 2 * np.pi / stream.sensor.num_columns
    a = rng.uniform(-amp_columns, amp_columns, stream.n_firings) * w
    ca, sa = np.cos(a)[:, None], np.sin(a)[:, None]
    x, y = stream.xyz[..., 0].astype(np.float64), stream.xyz[..., 1].astype(np.float64)
    xyz = stream.xyz.copy()
    xyz[..., 0] = (ca * x - sa * y).astype(np.float32)
    xyz[..., 1] = (sa * x + ca * y).astype(np.float32)
    return synth.Stream(xyz=xyz, intensity=stream.intensity, poses=stream.poses, sensor=stream.sensor, hit=stream.hit)


ROBOT_TF_TILTED = np.array([0.9961946980917455, 0.0, 0.08715574274765817, 1.2,
                            0.0, 1.0, 0.0, 0.1,
                            -0.08715574274765817, 0.0, 0.9961946980917455, 0.3], dtype=np.float64)


def build_case(name: str):
    if name == "s64_static":
        return synth.make_stream(720 * 3, seed=1234, sensor=_s64(720)), _kitti(720), None
    if name == "s64_translate":
        return synth.make_stream(720 * 3, seed=7, sensor=_s64(720), motion=Motion.translate()), _kitti(720), None
    if name == "s64_turn":
        return synth.make_stream(720 * 3, seed=8, sensor=_s64(720), motion=Motion.turn(10.0, 0.6)), _kitti(720), None
    if name == "s64_full_2200":
        return synth.make_stream(2200 * 2 + 300, seed=21, motion=Motion.translate()), _kitti(2200), None
    if name == "s64_forced_finish_ring":
        # unbroken wall ring: clusters exceed one rotation -> tree-width refusals, finished-root refusals, forced finish
        sc = SceneModel(n_objects=0, wall_radius=12.0, wall_gaps_deg=())
        return synth.make_stream(360 * 4, seed=5, sensor=_s64(360), scene=sc), _kitti(360), None
    if name == "s64_ring_with_objects":
        sc = SceneModel(n_objects=25, wall_radius=14.0, wall_gaps_deg=(), object_range=(3.0, 11.0))
        return synth.make_stream(360 * 4, seed=6, sensor=_s64(360), scene=sc, motion=Motion.translate(3.0)), _kitti(360), None
    if name == "s64_fog_and_ego":
        sc = SceneModel(n_objects=40, object_range=(1.5, 12.0), object_radius=(0.2, 0.6))
        cfg = _kitti(720, fog_filtering_enabled=1, fog_filtering_intensity_below=40, fog_filtering_distance_below=18.0,
                     fog_filtering_inclination_above=-0.2)
        return synth.make_stream(720 * 2 + 50, seed=9, sensor=_s64(720), scene=sc), cfg, None
    if name == "s64_counterclockwise":
        sen = _s64(720)
        sen.clockwise = False
        return synth.make_stream(720 * 2 + 50, seed=10, sensor=sen), _kitti(720, sensor_is_clockwise=0), None
    if name == "s64_every_2nd_column":
        return synth.make_stream(720 * 2 + 50, seed=11, sensor=_s64(720)), _kitti(720, cluster_point_trees_every_nth_column=2), None
    if name == "s64_no_early_stop":
        # stop_after_association disabled: many links per point -> exercises the exact serial association path
        return synth.make_stream(360 * 2 + 50, seed=12, sensor=_s64(360)), _kitti(360, stop_after_association_enabled=0), None
    if name == "s64_wide_window_global_kernel":
        # max_steps_in_row beyond the LDS window of k_assoc_lds: every stream runs the global-memory association kernel
        return synth.make_stream(720 * 2 + 50, seed=22, sensor=_s64(720)), _kitti(720, max_steps_in_row=31), None
    if name == "s64_min_steps_3":
        return synth.make_stream(720 * 2 + 50, seed=13, sensor=_s64(720)), _kitti(720, stop_after_association_min_steps=3), None
    if name == "s64_dropouts":
        sc = SceneModel(dropout=0.35, max_range=40.0)
        return synth.make_stream(720 * 2 + 50, seed=14, sensor=_s64(720), scene=sc, motion=Motion.translate()), _kitti(720), None
    if name == "s64_no_supplement_no_incl_ignore":
        cfg = _kitti(720, supplement_inclination_angle_for_nan_cells=0, ignore_points_with_too_big_inclination_angle_diff=0)
        return synth.make_stream(720 * 2 + 50, seed=15, sensor=_s64(720), scene=SceneModel(dropout=0.2)), cfg, None
    if name == "s64_robot_tf_tilted":
        cfg = _kitti(720)
        return synth.make_stream(720 * 2 + 50, seed=16, sensor=_s64(720), motion=Motion.turn(8.0, 0.3)), cfg, ROBOT_TF_TILTED
    if name == "s64_deep_lookback":
        # everything "flat" and every ground cell below a new obstacle "close": the downward fix-up of cc.cpp:513-535 relabels runs of up to
        # 62 rows, far beyond the 16 rows k_seg_scan keeps in LDS (it reads the rest from the staging plane)
        cfg = _kitti(360, max_slope=5.0, obstacle_because_next_certain_obstacle_max_dist_diff=1.0e6)
        sc = SceneModel(n_objects=30, object_range=(3.0, 15.0))
        return synth.make_stream(360 * 2 + 50, seed=31, sensor=_s64(360), scene=sc), cfg, None
    if name == "s128_offsets":
        return synth.make_stream(680 * 3, seed=17, sensor=_s128(680), start_column=20), _vls(680), None
    if name == "s128_no_offsets_translate":
        return synth.make_stream(680 * 3, seed=18, sensor=_s128(680, False), motion=Motion.translate()), _vls(680), None
    if name == "s128_full_1700":
        return synth.make_stream(1700 * 2 + 200, seed=19, sensor=SensorModel.s128(), start_column=40,
                                 motion=Motion.translate()), _vls(1700), None
    if name == "s96_offsets":
        # 96 rows: two rows per lane with a half-filled second register (lanes 32 - 63 own no second row), per-laser azimuth offsets
        sen = SensorModel(num_rows=96, num_columns=600, incl_top_deg=12.0, incl_bottom_deg=-25.0, azimuth_offsets_deg=(-3.1, -1.0, 1.0, 3.1))
        return synth.make_stream(600 * 3 + 40, seed=23, sensor=sen, start_column=15, motion=Motion.translate()), _vls(600), None
    if name == "s32_small_sensor":
        sen = SensorModel(num_rows=32, num_columns=512, incl_top_deg=10.0, incl_bottom_deg=-30.0)
        return synth.make_stream(512 * 3, seed=20, sensor=sen), _vls(512, max_distance=0.5), None
    # ---- jittered firing azimuths: collisions, shifts, refusals, skipped and revisited columns in the insertion
    if name == "j_s64_jitter":
        st = synth.make_stream(720 * 2 + 50, seed=51, sensor=_s64(720), motion=Motion.translate())
        return jitter_azimuth(st, 151, 0.8), _kitti(720), None
    if name == "j_s64_jitter_wide":
        st = synth.make_stream(360 * 3, seed=52, sensor=_s64(360), motion=Motion.turn(8.0, 0.4), scene=SceneModel(dropout=0.15))
        return jitter_azimuth(st, 152, 2.5), _kitti(360), ROBOT_TF_TILTED
    if name == "j_s128_offsets_jitter":
        st = synth.make_stream(680 * 2, seed=53, sensor=_s128(680), start_column=20, motion=Motion.translate())
        return jitter_azimuth(st, 153, 1.2), _vls(680), None
    # ---- ring-wrap cases: >= 12 rotations through the 10-rotation ring (cc.cpp:17), small columns-per-rotation so the oracle stays fast
    if name == "w_s64_240x13":
        sc = SceneModel(n_objects=40, object_range=(4.0, 30.0))
        return synth.make_stream(240 * 13, seed=41, sensor=_s64(240), scene=sc, motion=Motion.translate()), _kitti(240), None
    if name == "w_s64_360x12_turn":
        return synth.make_stream(360 * 12 + 100, seed=42, sensor=_s64(360), motion=Motion.turn(8.0, 0.5)), _kitti(360), ROBOT_TF_TILTED
    if name == "w_s64_ring_wall_240x12":
        # unbroken wall ring + objects: forced finishes (cluster wider than a rotation) on every rotation, across the ring wrap
        sc = SceneModel(n_objects=25, wall_radius=14.0, wall_gaps_deg=(), object_range=(3.0, 11.0))
        return synth.make_stream(240 * 12 + 60, seed=43, sensor=_s64(240), scene=sc, motion=Motion.translate(3.0)), _kitti(240), None
    if name == "w_s128_offsets_340x12":
        return synth.make_stream(340 * 12 + 80, seed=44, sensor=_s128(340), start_column=12, motion=Motion.translate()), _vls(340), None
    if name == "w_s32_256x12":
        sen = SensorModel(num_rows=32, num_columns=256, incl_top_deg=10.0, incl_bottom_deg=-30.0)
        return synth.make_stream(256 * 12 + 40, seed=45, sensor=sen, motion=Motion.translate()), _vls(256, max_distance=0.5), None
    # ---- gaps in slanted surfaces: a tree that has gone quiet for about as many columns as its points' max angle difference reaches is met
    # again by a later, nearer point (large cylinders seen off-centre: the range changes quickly with the azimuth; every few dozen columns the
    # obstacle returns of 4 - 15 consecutive firings are dropped). The reference then refuses attaches to trees whose cluster finished meanwhile
    # (cc.cpp:658) or keeps a cluster alive through a link: the exceptions the batch-parallel association kernel has to detect
    if name in ("x_s64_slanted_gaps", "x_s64_slanted_gaps_far"):
        far = name.endswith("far")
        sc = SceneModel(n_objects=14, object_range=(14.0, 40.0) if far else (7.0, 24.0), object_radius=(3.0, 11.0), object_top_z=2.5,
                        wall_radius=30.0 if far else 0.0, wall_gaps_deg=((10.0, 50.0), (200.0, 260.0)), dropout=0.05)
        st = synth.make_stream(720 * 3, seed=61 if far else 62, sensor=_s64(720), scene=sc, motion=Motion.translate(6.0))
        xyz = st.xyz.copy()
        rng = np.random.default_rng(161)
        k = 0
        while k < st.n_firings:
            k += int(rng.integers(9, 40))
            g = int(rng.integers(3, 16))
            sel = st.hit[k:k + g] > 1  # obstacle returns only: the ground keeps the columns (and the insertion) regular
            xyz[k:k + g][sel] = np.nan
            k += g
        return synth.Stream(xyz=xyz, intensity=st.intensity, poses=st.poses, sensor=st.sensor, hit=st.hit), _kitti(720), None
    if name == "x_s64_refused_attach":
        # Hand-made: the attach the reference REFUSES because the candidate's tree belongs to a finished cluster (cc.cpp:658). A tree lives until a
        # column's smallest AZIMUTH passes its largest (azimuth + max angle difference); a later point accepts a candidate when their 3-D distance is
        # below max_distance. For steep lasers the 3-D angle is smaller than the azimuth difference (cos(inclination) < 1), so a post seen by the rows
        # at -21..-23 degrees at 3 m is still within 0.5 m of an equal post 20 columns later although its tree was finished a column earlier — provided
        # the first post sits at the very start of its column and the second one too. 720 columns, static sensor, unbroken wall (every row returns).
        sen = _s64(720)
        sc = SceneModel(n_objects=0, wall_radius=40.0, wall_gaps_deg=(), range_noise=0.0, dropout=0.0)
        st = synth.make_stream(720 * 2 + 100, seed=81, sensor=sen, scene=sc)
        xyz = st.xyz.astype(np.float64)
        w = 2 * np.pi / 720
        incl = np.deg2rad(np.linspace(sen.incl_top_deg, sen.incl_bottom_deg, 64))
        frac = np.full(st.n_firings, 0.5)
        posts = []
        k = 60
        gaps = [20, 20, 19, 20, 18, 20, 20, 17, 20, 20, 20, 16, 20, 20]
        i = 0
        while k + 25 < st.n_firings:
            gap = gaps[i % len(gaps)]
            posts.append((k, k + gap))
            frac[k] = 0.02
            frac[k + 1:k + gap] = 0.5
            frac[k + gap] = 0.01
            k += gap + 31 + (i % 5)
            i += 1
        # rotate every firing about z so that it sits at (k + frac) * w instead of (k + 0.5) * w (clockwise sensor: azimuth = pi - that)
        rot = -(frac - 0.5) * w
        ca, sa = np.cos(rot)[:, None], np.sin(rot)[:, None]
        x, y = xyz[..., 0].copy(), xyz[..., 1].copy()
        xyz[..., 0] = ca * x - sa * y
        xyz[..., 1] = sa * x + ca * y
        rows = np.arange(54, 59)
        for a, b in posts:
            for kk in (a, b):
                az = np.pi - (kk + frac[kk]) * w
                rng_ = 3.0
                xyz[kk, rows, 0] = rng_ * np.cos(incl[rows]) * np.cos(az)
                xyz[kk, rows, 1] = rng_ * np.cos(incl[rows]) * np.sin(az)
                xyz[kk, rows, 2] = rng_ * np.sin(incl[rows])
        return synth.Stream(xyz=xyz.astype(np.float32), intensity=st.intensity, poses=st.poses, sensor=sen, hit=st.hit), _kitti(720), None
    if name.startswith("x_s64_near_jitter_gaps"):
        # The same close to the sensor with jittered firing azimuths: a point reaches back ceil(max angle difference / column width) columns, a
        # tree lives until a column's smallest azimuth passes its largest azimuth + max angle difference — with every firing at its own fraction
        # of a column and a nearer point looking back at a farther one the two can disagree by a column (only below ~6 m at 720 columns).
        seed = int(name.rsplit("_", 1)[1]) if name[-1].isdigit() else 0
        sc = SceneModel(n_objects=26, object_range=(2.2, 6.5), object_radius=(0.4, 3.0), object_top_z=1.5, wall_radius=0.0, dropout=0.04)
        st = synth.make_stream(720 * 3, seed=70 + seed, sensor=_s64(720), scene=sc, motion=Motion.translate(4.0))
        xyz = st.xyz.copy()
        rng = np.random.default_rng(170 + seed)
        k = 0
        while k < st.n_firings:
            k += int(rng.integers(6, 30))
            g = int(rng.integers(2, 19))
            sel = st.hit[k:k + g] > 1
            xyz[k:k + g][sel] = np.nan
            k += g
        st = synth.Stream(xyz=xyz, intensity=st.intensity, poses=st.poses, sensor=st.sensor, hit=st.hit)
        return jitter_azimuth(st, 270 + seed, 0.9), _kitti(720), None
    # ---- vegetation-like scenes: what leaves the batch-parallel association's fast path on natural data --------------------------------
    if name == "c_s64_sparse_clutter":
        # isolated returns at 3 .. 30 m all around: more unfinished trees side by side than the kernel has lanes for (AB_BAIL_TREES)
        return synth.make_stream(2200 * 2 + 300, seed=77, sensor=_s64(2200), scene=SceneModel.sparse_clutter(0.1), motion=Motion.translate()), _kitti(2200), None
    if name == "c_s64_near_clutter":
        # a shell of returns at 2 .. 8 m all around: a tree that may reach the one-rotation limits (AB_BAIL_ROTATION)
        return synth.make_stream(2200 * 2 + 300, seed=77, sensor=_s64(2200), scene=SceneModel.near_clutter(0.1), motion=Motion.translate()), _kitti(2200), None
    if name == "c_s64_mixed_clutter":
        return synth.make_stream(2200 * 2 + 300, seed=78, sensor=_s64(2200), scene=SceneModel.cluttered(0.1), motion=Motion.turn(8.0, 0.15)), _kitti(2200), None
    if name == "c_s128_sparse_clutter":
        return synth.make_stream(1700 + 600, seed=79, sensor=_s128(1700), scene=SceneModel.sparse_clutter(0.08), motion=Motion.translate()), _vls(1700), None
    # ---- small variants kept as committed golden fixtures -------------------------------------------------------
    if name == "g_s64_translate":
        return synth.make_stream(800, seed=31, sensor=_s64(360), motion=Motion.translate()), _kitti(360), None
    if name == "g_s64_forced_finish_ring":
        sc = SceneModel(n_objects=0, wall_radius=12.0, wall_gaps_deg=())
        return synth.make_stream(240 * 4, seed=32, sensor=_s64(240), scene=sc), _kitti(240), None
    if name == "g_s128_offsets":
        return synth.make_stream(816, seed=33, sensor=_s128(340), start_column=12, motion=Motion.turn(6.0, 0.4)), _vls(340), None
    if name == "g_s64_fog_and_ego":
        sc = SceneModel(n_objects=40, object_range=(1.5, 12.0), object_radius=(0.2, 0.6), dropout=0.1)
        cfg = _kitti(360, fog_filtering_enabled=1, fog_filtering_intensity_below=40, fog_filtering_distance_below=18.0,
                     fog_filtering_inclination_above=-0.2)
        return synth.make_stream(800, seed=34, sensor=_s64(360), scene=sc), cfg, ROBOT_TF_TILTED
    raise KeyError(name)


ALL_CASES = ["s64_static", "s64_translate", "s64_turn", "s64_full_2200", "s64_forced_finish_ring", "s64_ring_with_objects",
             "s64_fog_and_ego", "s64_counterclockwise", "s64_every_2nd_column", "s64_no_early_stop", "s64_min_steps_3",
             "s64_wide_window_global_kernel", "s64_dropouts", "s64_no_supplement_no_incl_ignore", "s64_robot_tf_tilted", "s128_offsets",
             "s128_no_offsets_translate", "s128_full_1700", "s96_offsets", "s32_small_sensor", "j_s64_jitter", "j_s64_jitter_wide", "j_s128_offsets_jitter"]

EXCEPTION_CASES = ["x_s64_slanted_gaps", "x_s64_slanted_gaps_far", "x_s64_near_jitter_gaps_3", "x_s64_refused_attach"]

CLUTTER_CASES = ["c_s64_sparse_clutter", "c_s64_near_clutter", "c_s64_mixed_clutter", "c_s128_sparse_clutter"]

RING_WRAP_CASES = ["w_s64_240x13", "w_s64_360x12_turn", "w_s64_ring_wall_240x12", "w_s128_offsets_340x12", "w_s32_256x12"]

# cases stored as golden fixtures under tests/golden/ (inputs + expected outputs)
GOLDEN_CASES = ["g_s64_translate", "g_s64_forced_finish_ring", "g_s128_offsets", "g_s64_fog_and_ego"]
