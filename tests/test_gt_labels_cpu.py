"""CPU: the oracle of generateEuclideanClusteringLabels (oracle/gt_oracle.cpp, the PCL region-growing loop restated) against an
independent formulation: brute-force float32 pair distances + scipy connected components, numbered by first point."""
import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components

from oracle import pyoracle as orc

GROUND = (60, 40, 44, 48, 49, 72, 0)


def scene(seed, n=2500, spread=12.0):
    rng = np.random.default_rng(seed)
    centers = rng.uniform(-spread, spread, (40, 3)).astype(np.float32)
    which = rng.integers(0, 40, n)
    pts = centers[which] + rng.normal(0, 0.45, (n, 3)).astype(np.float32)
    sem_of = rng.choice(np.array([10, 30, 40, 50, 70, 0, 72, 80], dtype=np.uint16), 40)
    sem = sem_of[which]
    inst = (which % 3).astype(np.uint16)
    far = rng.random(n) < 0.05
    pts[far] = rng.uniform(-60, 60, (far.sum(), 3)).astype(np.float32)
    return np.concatenate([pts, rng.random((n, 1)).astype(np.float32)], axis=1).astype(np.float32), sem, inst


def brute_force(pts, sem, inst):
    x = pts[:, :3].astype(np.float32)
    d = (x[:, None, :] - x[None, :, :]).astype(np.float32)
    sq = (d * d).astype(np.float32)
    d2 = ((sq[..., 0] + sq[..., 1]).astype(np.float32) + sq[..., 2]).astype(np.float32)
    same = (sem[:, None] == sem[None, :]) & (inst[:, None] == inst[None, :])
    adj = (d2 < np.float32(1.0)) & same
    n = len(pts)
    ncomp, comp = connected_components(coo_matrix(adj), directed=False)
    sizes = np.bincount(comp, minlength=ncomp)
    first = np.full(ncomp, n)
    np.minimum.at(first, comp, np.arange(n))
    kept = [c for c in np.argsort(first) if 10 <= sizes[c] <= 300000]
    number = {c: k + 1 for k, c in enumerate(kept)}
    out = np.array([0 if s in GROUND else number.get(c, 0) for s, c in zip(sem, comp)], dtype=np.uint16)
    return out, len(kept)


def test_oracle_equals_brute_force_components():
    for seed in range(4):
        pts, sem, inst = scene(seed)
        got, nc = orc.generate_euclidean_labels(pts, sem, inst)
        want, nk = brute_force(pts, sem, inst)
        assert nc == nk and nc > 5
        assert np.array_equal(got, want)
        assert (got[np.isin(sem, GROUND)] == 0).all() and got.max() <= nc


def test_oracle_edge_cases():
    z = np.zeros(0, np.uint16)
    got, nc = orc.generate_euclidean_labels(np.zeros((0, 4), np.float32), z, z)
    assert nc == 0 and got.shape == (0,)
    # a chain of 12 points 0.9 m apart is one cluster; 1.0 m apart (not < 1) is twelve singletons; NaN points never join
    for step, expect in ((0.9, 1), (1.0, 0)):
        pts = np.zeros((12, 4), np.float32)
        pts[:, 0] = np.arange(12, dtype=np.float32) * np.float32(step)
        got, nc = orc.generate_euclidean_labels(pts, np.full(12, 10, np.uint16), np.zeros(12, np.uint16))
        assert nc == expect and (got == expect).all()
    pts = np.zeros((14, 4), np.float32)
    pts[:, 1] = np.arange(14, dtype=np.float32) * np.float32(0.3)
    pts[3, 0] = np.nan
    sem = np.full(14, 50, np.uint16)
    sem[7] = 51                                   # a different label splits the chain only if the gap reaches 1 m: 0.3-m spacing bridges it
    got, nc = orc.generate_euclidean_labels(pts, sem, np.zeros(14, np.uint16))
    assert nc == 1 and got[3] == 0 and got[7] == 0 and (np.delete(got, [3, 7]) == 1).all()
