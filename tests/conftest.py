import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def wal70():
    import json

    import numpy as np

    ids = json.load(open(os.path.join(GOLDEN, "wal70_ids.json")))
    top = json.load(open(os.path.join(GOLDEN, "wal70_top5.json")))
    V = np.fromfile(os.path.join(GOLDEN, "wal70_vectors.f32"), dtype="<f4").reshape(70, 384)
    return {"ids": ids["ids"], "metadatas": ids["metadatas"], "log": ids["log"], "vectors": V,
            "top_rows": np.array(top["rows"]), "top_cos": np.array(top["cos"])}
