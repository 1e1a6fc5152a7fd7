"""Real-data acceptance hook (BASELINE.json configs[0] / [4]; /root/reference/README.md:213-245): where a SemanticKITTI root is mounted
($SEMANTIC_KITTI_ROOT, the directory holding sequences/), the listed sequences are replayed through the HIP path like the reference's
kitti_demo and the per-sequence quality table must equal the one the reference publishes, at the printed two decimals. Skips without the
dataset (it is not in this image). $SEMANTIC_KITTI_SEQUENCES = "4" or "0,1,2" restricts the run (default: every train sequence found)."""
import os

import pytest

ROOT = os.environ.get("SEMANTIC_KITTI_ROOT", "")

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (ROOT and os.path.isdir(os.path.join(ROOT, "sequences"))), reason="no SemanticKITTI root in $SEMANTIC_KITTI_ROOT")]


def test_train_sequences_reproduce_the_published_tables():
    from continuous_clustering_amd import acceptance
    want = os.environ.get("SEMANTIC_KITTI_SEQUENCES", "")
    seqs = [int(s) for s in want.split(",") if s.strip()] or list(acceptance.TRAIN_SEQUENCES)
    seqs = acceptance.available_sequences(ROOT, seqs)
    assert seqs, "no complete sequence folder found"
    res = acceptance.run(ROOT, seqs)
    bad = {k: {m: (c["got"], c["want"]) for m, c in row.items() if m != "frames" and c["ok"] is False} for k, row in res["rows"].items()}
    bad = {k: v for k, v in bad.items() if v}
    assert res["cells_checked"] == 6 * len(res["rows"]) and not bad, bad
    for s in seqs:
        assert res["rows"][str(s)]["frames"] == acceptance.TRAIN_FRAMES[s]
