"""TEST ONLY. A stand-in with the Engine methods bench.py's throughput leg calls, so that the LAUNCHER and the aggregation over ranks
(`python bench.py --gpus N` starting N ranks, barriers, max-over-ranks time, summed counts, strong split dealing) can be tested on CPU over gloo
(tests/test_bench_launcher.py). It clusters nothing: a "step" sleeps and counts the cells it was handed. Never used by the product or by a
measurement (bench.py --stub-engine marks its line as such)."""
import time


class StubEngine:
    STEP_SECONDS = 0.01

    def __init__(self, cfg, num_rows, num_streams, rank=0):
        self.R, self.S, self.rank = num_rows, num_streams, rank
        self.cells = 0
        self.batches = 0

    def record_events(self, enable):
        pass

    def set_option(self, name, value):
        pass

    def add_firings_device(self, n, d_xyz, d_intensity, d_poses):
        assert tuple(d_xyz.shape) == (self.S, n, self.R, 3), (tuple(d_xyz.shape), self.S, n, self.R)
        time.sleep(self.STEP_SECONDS * (1 + self.rank))  # rank 1 is slower: the job's time must be the slowest rank's
        self.cells += self.S * n * self.R
        self.batches += 1

    def sync(self):
        return 0

    def last_error(self):
        return ""

    def totals(self):
        return {"cells_published": self.cells, "clusters_finished": 7 * self.batches, "firings_consumed": 0, "serial_columns": 0}

    def enable_timing(self, enable=True):
        if enable:
            self.batches = 0

    def kernel_times(self):
        d = {k: 1.0 * self.batches for k in ("prep_ms", "insert_ms", "segment_ms", "scan_ms", "assoc_lds_ms", "assoc_global_ms", "publish_ms")}
        d["batches"] = self.batches
        return d

    def batch_counters(self):
        return {"batch_columns": 0, "batch_bails": 0, "bail_reasons": [0] * 8}

    def close(self):
        pass
