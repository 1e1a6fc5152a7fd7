"""ctypes mirror of include/cc_hip.h (struct layouts, constants). Pure declarations, no compute."""
from __future__ import annotations

import ctypes as C

import numpy as np

CC_OK = 0
CC_ERR_INVALID_ARGUMENT = 1
CC_ERR_HIP = 2
CC_ERR_NO_DEVICE = 3
CC_ERR_FIRING_SIZE = 4
CC_ERR_NO_ROBOT_TRANSFORM = 5
CC_ERR_RING_OVERRUN = 6
CC_ERR_BOOKKEEPING = 7
CC_ERR_CAPACITY = 8
CC_ERR_NEGATIVE_COLUMN = 9

GP_UNKNOWN, GP_GROUND, GP_OBSTACLE, GP_EGO_VEHICLE, GP_FOG = 143, 54, 119, 85, 71

EV_GROUND_COLUMN, EV_CLUSTER, EV_PUBLISH_COLUMNS = 1, 2, 3


class Config(C.Structure):
    """cc_config == continuous_clustering::Configuration (continuous_clustering.hpp:24-87)."""
    _fields_ = [
        ("is_single_threaded", C.c_int32),
        ("sensor_is_clockwise", C.c_int32),
        ("num_columns", C.c_int32),
        ("supplement_inclination_angle_for_nan_cells", C.c_int32),
        ("max_slope", C.c_float),
        ("first_ring_as_ground_max_allowed_z_diff", C.c_float),
        ("first_ring_as_ground_min_allowed_z_diff", C.c_float),
        ("last_ground_point_slope_higher_than", C.c_float),
        ("last_ground_point_distance_smaller_than", C.c_float),
        ("ground_because_close_to_last_certain_ground_max_z_diff", C.c_float),
        ("ground_because_close_to_last_certain_ground_max_dist_diff", C.c_float),
        ("obstacle_because_next_certain_obstacle_max_dist_diff", C.c_float),
        ("use_terrain", C.c_int32),
        ("terrain_max_allowed_z_diff", C.c_float),
        ("height_ref_to_maximum_", C.c_float),
        ("height_ref_to_ground_", C.c_float),
        ("length_ref_to_front_end_", C.c_float),
        ("length_ref_to_rear_end_", C.c_float),
        ("width_ref_to_left_mirror_", C.c_float),
        ("width_ref_to_right_mirror_", C.c_float),
        ("fog_filtering_enabled", C.c_int32),
        ("fog_filtering_intensity_below", C.c_int32),
        ("fog_filtering_distance_below", C.c_float),
        ("fog_filtering_inclination_above", C.c_float),
        ("max_distance", C.c_float),
        ("max_steps_in_row", C.c_int32),
        ("max_steps_in_column", C.c_int32),
        ("stop_after_association_enabled", C.c_int32),
        ("stop_after_association_min_steps", C.c_int32),
        ("ignore_points_in_chessboard_pattern", C.c_int32),
        ("ignore_points_with_too_big_inclination_angle_diff", C.c_int32),
        ("use_last_point_for_cluster_stamp", C.c_int32),
        ("cluster_point_trees_every_nth_column", C.c_int32),
    ]

    @staticmethod
    def default() -> "Config":
        """Library defaults, continuous_clustering.hpp:24-79."""
        c = Config()
        c.is_single_threaded = 0
        c.sensor_is_clockwise = 1
        c.num_columns = 1700
        c.supplement_inclination_angle_for_nan_cells = 1
        c.max_slope = 0.2
        c.first_ring_as_ground_max_allowed_z_diff = 0.4
        c.first_ring_as_ground_min_allowed_z_diff = -0.4
        c.last_ground_point_slope_higher_than = -0.1
        c.last_ground_point_distance_smaller_than = 5.0
        c.ground_because_close_to_last_certain_ground_max_z_diff = 0.4
        c.ground_because_close_to_last_certain_ground_max_dist_diff = 2.0
        c.obstacle_because_next_certain_obstacle_max_dist_diff = 0.3
        c.use_terrain = 0
        c.terrain_max_allowed_z_diff = 0.4
        c.fog_filtering_enabled = 0
        c.fog_filtering_intensity_below = 2
        c.fog_filtering_distance_below = 18.0
        c.fog_filtering_inclination_above = -0.06
        c.max_distance = 0.7
        c.max_steps_in_row = 20
        c.max_steps_in_column = 20
        c.stop_after_association_enabled = 1
        c.stop_after_association_min_steps = 1
        c.ignore_points_in_chessboard_pattern = 1
        c.ignore_points_with_too_big_inclination_angle_diff = 1
        c.use_last_point_for_cluster_stamp = 0
        c.cluster_point_trees_every_nth_column = 1
        return c

    @staticmethod
    def kitti() -> "Config":
        """The KITTI parameters of src/tools/kitti_demo.cpp:279-294."""
        c = Config.default()
        c.is_single_threaded = 1
        c.num_columns = 2200
        c.ignore_points_in_chessboard_pattern = 0
        c.max_distance = 0.5
        c.height_ref_to_maximum_ = 0.5
        c.height_ref_to_ground_ = -1.7
        c.length_ref_to_front_end_ = 3.0
        c.length_ref_to_rear_end_ = -3.0
        c.width_ref_to_left_mirror_ = 1.5
        c.width_ref_to_right_mirror_ = -1.5
        return c

    @staticmethod
    def vls128() -> "Config":
        """Library defaults with the VLS-128 column count (launch/sensor_vls128_roof.launch:22) and an ego box."""
        c = Config.default()
        c.is_single_threaded = 1
        c.num_columns = 1700
        c.height_ref_to_maximum_ = 0.5
        c.height_ref_to_ground_ = -1.7
        c.length_ref_to_front_end_ = 3.0
        c.length_ref_to_rear_end_ = -3.0
        c.width_ref_to_left_mirror_ = 1.5
        c.width_ref_to_right_mirror_ = -1.5
        return c

    def copy(self) -> "Config":
        c = Config()
        C.memmove(C.byref(c), C.byref(self), C.sizeof(Config))
        return c


class Event(C.Structure):
    _fields_ = [("type", C.c_int32), ("stream", C.c_int32), ("a", C.c_int64), ("b", C.c_int64),
                ("c", C.c_uint32), ("d", C.c_uint32), ("column", C.c_int64)]


EVENT_DTYPE = np.dtype([("type", "<i4"), ("stream", "<i4"), ("a", "<i8"), ("b", "<i8"), ("c", "<u4"), ("d", "<u4"),
                        ("column", "<i8")])
assert EVENT_DTYPE.itemsize == C.sizeof(Event)


class StreamState(C.Structure):
    _fields_ = [
        ("num_rows", C.c_int32), ("num_columns", C.c_int32), ("ring_buffer_max_columns", C.c_int32),
        ("reset_required", C.c_int32),
        ("ring_buffer_start_global_column_index", C.c_int64), ("ring_buffer_end_global_column_index", C.c_int64),
        ("first_unfinished_global_column_index", C.c_int64), ("first_unpublished_global_column_index", C.c_int64),
        ("cluster_counter", C.c_uint64), ("firings_consumed", C.c_uint64), ("cells_published", C.c_uint64),
        ("clusters_finished", C.c_uint64),
        ("error", C.c_int32), ("n_unfinished_trees", C.c_int32), ("error_a", C.c_int64), ("error_b", C.c_int64),
    ]


class ColumnView(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("y", C.c_void_p), ("z", C.c_void_p), ("distance", C.c_void_p),
        ("inclination_angle", C.c_void_p), ("continuous_azimuth_angle", C.c_void_p),
        ("global_column_index", C.c_void_p), ("source_firing", C.c_void_p),
        ("ground_point_label", C.c_void_p), ("debug_ground_point_label", C.c_void_p), ("is_ignored", C.c_void_p),
        ("id", C.c_void_p), ("tree_root_global_column", C.c_void_p), ("tree_root_row", C.c_void_p),
        ("finished_at_continuous_azimuth_angle", C.c_void_p), ("tree_num_points", C.c_void_p), ("cluster_width", C.c_void_p),
        ("number_of_child_points", C.c_void_p), ("number_of_visited_neighbors", C.c_void_p),
        ("belongs_to_finished_cluster", C.c_void_p), ("tree_parent_global_column", C.c_void_p), ("tree_parent_row", C.c_void_p),
    ]


COLUMN_FIELDS = {
    "x": np.float32, "y": np.float32, "z": np.float32, "distance": np.float32, "inclination_angle": np.float32,
    "continuous_azimuth_angle": np.float64, "global_column_index": np.int64, "source_firing": np.int64,
    "ground_point_label": np.uint8, "debug_ground_point_label": np.uint8, "is_ignored": np.uint8, "id": np.uint64,
    "tree_root_global_column": np.int64, "tree_root_row": np.int32,
    "finished_at_continuous_azimuth_angle": np.float64, "tree_num_points": np.uint32, "cluster_width": np.uint32,
    "number_of_child_points": np.uint32, "number_of_visited_neighbors": np.int32, "belongs_to_finished_cluster": np.uint8,
    "tree_parent_global_column": np.int64, "tree_parent_row": np.int32,
}


def make_column_view(n_cols: int, n_rows: int, fields=None):
    """Allocate numpy arrays [n_cols, n_rows] for the requested fields and a ColumnView pointing at them."""
    fields = list(COLUMN_FIELDS) if fields is None else list(fields)
    arrays = {f: np.zeros((n_cols, n_rows), dtype=COLUMN_FIELDS[f]) for f in fields}
    v = ColumnView()
    for f, a in arrays.items():
        setattr(v, f, a.ctypes.data)
    return v, arrays


def state_to_dict(s: StreamState) -> dict:
    return {name: getattr(s, name) for name, _ in StreamState._fields_}
