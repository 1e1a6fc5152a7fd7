import numpy as np

from continuous_clustering_amd import synth


def test_stream_shapes_and_determinism():
    sen = synth.SensorModel(num_rows=64, num_columns=360)
    a = synth.make_stream(400, seed=3, sensor=sen, motion=synth.Motion.turn())
    b = synth.make_stream(400, seed=3, sensor=sen, motion=synth.Motion.turn())
    assert a.xyz.shape == (400, 64, 3) and a.xyz.dtype == np.float32
    assert a.intensity.shape == (400, 64) and a.poses.shape == (400, 12)
    assert np.array_equal(a.xyz.view(np.uint32), b.xyz.view(np.uint32)) and np.array_equal(a.poses, b.poses)
    c = synth.make_stream(400, seed=4, sensor=sen)
    assert not np.array_equal(a.xyz.view(np.uint32), c.xyz.view(np.uint32))
    # poses are rigid: R^T R = I
    R = a.poses.reshape(-1, 3, 4)[:, :, :3]
    assert np.allclose(np.einsum("nij,nkj->nik", R, R), np.eye(3), atol=1e-12)
    # firing k looks at column k of the rotation (clockwise sensor, cc.cpp:146-151)
    az = np.arctan2(a.xyz[..., 1], a.xyz[..., 0])
    col = ((np.pi - az) / (2 * np.pi / 360)).astype(int)
    valid = ~np.isnan(a.xyz[..., 0])
    expect = (np.arange(400)[:, None] % 360) * np.ones((1, 64), dtype=int)
    assert (np.abs(col[valid] - expect[valid]) <= 1).mean() > 0.999
