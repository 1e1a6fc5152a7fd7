"""GPU: the HIP path reproduces the committed golden fixtures (no oracle involved at run time)."""
import numpy as np
import pytest

import cases
import util
from continuous_clustering_amd import capi
from test_oracle_golden import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", cases.GOLDEN_CASES)
def test_engine_matches_golden(name):
    from continuous_clustering_amd import Engine, IDENTITY_TF
    z, cfg, tf = load_golden(name)
    R = int(z["num_rows"])
    e = Engine(cfg, R, 1, 0, IDENTITY_TF if tf is None else tf)
    gold_ev = z["events"]
    out_fields = [k[4:] for k in z.files if k.startswith("out_")]
    first = int(z["first_column"])
    n = z["xyz"].shape[0]
    pos = 0
    chunk = cfg.num_columns
    for f in range(0, n, chunk):
        assert e.add_firings(z["xyz"][f:f + chunk], z["intensity"][f:f + chunk], z["poses"][f:f + chunk]) == 0
        ev = e.drain_events()
        ref = gold_ev[pos:pos + len(ev)]
        assert len(ref) == len(ev)
        for fld in ("type", "a", "b", "c", "d", "column"):
            assert np.array_equal(ev[fld], ref[fld]), fld
        pos += len(ev)
        pub = ev[(ev["type"] == capi.EV_PUBLISH_COLUMNS) & (ev["b"] >= ev["a"])]
        if len(pub):
            lo, hi = int(pub["a"].min()), int(pub["b"].max())
            cols = e.read_columns(lo, hi, fields=out_fields)
            for fld in out_fields:
                g = z["out_" + fld][lo - first:hi - first + 1]
                a = cols[fld]
                if a.dtype.kind == "f":
                    util.assert_float_equal(fld, a, g)
                else:
                    assert np.array_equal(a.astype(np.int64), g.astype(np.int64)), fld
    assert pos == len(gold_ev)
    st = e.state()
    assert [st[k] for k in util.STATE_FIELDS] == list(z["state"])
