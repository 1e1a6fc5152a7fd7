"""CPU: the oracle reproduces the committed golden fixtures (tests/golden/*.npz, see make_golden.py for provenance)."""
import os

import numpy as np
import pytest

import cases
import util
from continuous_clustering_amd import capi

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    cfg = capi.Config.from_buffer_copy(z["config"].tobytes())
    tf = z["robot_tf"] if z["robot_tf"].size else None
    return z, cfg, tf


@pytest.mark.parametrize("name", cases.GOLDEN_CASES)
def test_oracle_matches_golden(name, oracle_lib):
    from oracle.pyoracle import Oracle, IDENTITY_TF
    z, cfg, tf = load_golden(name)
    o = Oracle(cfg, int(z["num_rows"]), IDENTITY_TF if tf is None else tf)
    assert o.add_firings(z["xyz"], z["intensity"], z["poses"]) == 0
    ev = o.drain_events()
    gold = z["events"]
    assert len(ev) == len(gold)
    for f in ("type", "a", "b", "c", "d", "column"):
        assert np.array_equal(ev[f], gold[f]), f
    frm, to = o.published_range()
    assert frm == int(z["first_column"])
    out_fields = [k[4:] for k in z.files if k.startswith("out_")]
    cols = o.read_published(frm, to, out_fields)
    for f in out_fields:
        g = z["out_" + f]
        a = cols[f]
        if a.dtype.kind == "f":
            util.assert_float_equal(f, a, g)
        else:
            assert np.array_equal(a.astype(np.int64), g.astype(np.int64)), f
    st = o.state()
    assert [st[k] for k in util.STATE_FIELDS] == list(z["state"])


def test_golden_generation_is_reproducible():
    """The synthetic generator that made the fixtures is deterministic: regenerated inputs equal the stored ones."""
    for name in cases.GOLDEN_CASES[:2]:
        z, cfg, tf = load_golden(name)
        stream, cfg2, tf2 = cases.build_case(name)
        assert np.array_equal(stream.xyz.view(np.uint32), z["xyz"].view(np.uint32))
        assert np.array_equal(stream.intensity, z["intensity"])
        assert np.array_equal(stream.poses, z["poses"])
        assert bytes(cfg2) == bytes(cfg)
