"""Writes tests/golden/semantickitti_readme_tables.json: the quality tables the reference publishes for its own path on the SemanticKITTI train
sequences (/root/reference/README.md:213-245, obtained at the reference's commit fa3c53b), as printed — two decimals, mu / sigma. They are the
only known answers the reference holds for this path and need the dataset, so they are used by the acceptance hook
(continuous_clustering_amd/acceptance.py, tests/test_gpu_semantickitti.py, bench.py --kitti-root), which skips where no dataset is mounted.

    python tests/golden/make_readme_tables.py            # in the build container, where /root/reference exists
"""
import json
import os
import re

README = "/root/reference/README.md"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "semantickitti_readme_tables.json")


def main():
    lines = open(README).read().splitlines()
    tables = {}
    section = None
    for ln in lines:
        if ln.startswith("### 1.1."):
            section = ("use", "ose")
        elif ln.startswith("### 1.2."):
            section = ("recall", "precision", "f1", "accuracy")
        elif ln.startswith("## 2."):
            section = None
        if not section or not ln.startswith("|"):
            continue
        cells = [c.strip() for c in ln.strip().strip("|").split("|")]
        name = cells[0]
        if name.startswith("All (**Ours**)"):
            key = "all"
        elif re.fullmatch(r"\d+", name):
            key = str(int(name))
        else:
            continue
        row = tables.setdefault(key, {})
        for metric, cell in zip(section, cells[1:]):
            mu, sigma = [v.strip().strip("*") for v in cell.split("/")]
            row[metric] = [mu, sigma]
    assert len(tables) == 12 and all(len(r) == 6 for r in tables.values()), tables
    json.dump({"source": "UniBwTAS/continuous_clustering README.md:213-245 (commit fa3c53b), values as printed", "tables": tables}, open(OUT, "w"), indent=1, sort_keys=True)
    print(OUT)


if __name__ == "__main__":
    main()
