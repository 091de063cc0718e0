import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.join(ROOT, "tests")
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Golden vectors produced by the reference itself (tests/golden/make_golden.py)."""
    import json

    import numpy as np
    import torch

    model = np.load(os.path.join(GOLDEN, "small_model.npz"))
    cases = np.load(os.path.join(GOLDEN, "small_cases.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "meta.json")))
    dec = {k[4:]: torch.from_numpy(model[k]) for k in model.files if k.startswith("dec/")}
    post = {k[5:]: torch.from_numpy(model[k]) for k in model.files if k.startswith("post/")}
    enc = {k[4:]: torch.from_numpy(model[k]) for k in model.files if k.startswith("enc/")}
    return {
        "dec": dec,
        "post": post,
        "enc": enc,
        "ids": torch.from_numpy(model["ids"]),
        "memory": torch.from_numpy(model["memory"]),
        "lengths": torch.from_numpy(model["lengths"]),
        "cases": {k: torch.from_numpy(cases[k]) for k in cases.files},
        "meta": meta,
    }


@pytest.fixture(scope="session")
def golden_taco2():
    """Golden vectors of Taco2DecoderCell (r=2) + MelPostnet2 from the reference (make_golden_taco2.py)."""
    import json

    import numpy as np
    import torch

    model = np.load(os.path.join(GOLDEN, "taco2_model.npz"))
    cases = np.load(os.path.join(GOLDEN, "taco2_cases.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "taco2_meta.json")))
    return {
        "dec": {k[4:]: torch.from_numpy(model[k]) for k in model.files if k.startswith("dec/")},
        "post": {k[5:]: torch.from_numpy(model[k]) for k in model.files if k.startswith("post/")},
        "memory": torch.from_numpy(model["memory"]),
        "cases": {k: torch.from_numpy(cases[k]) for k in cases.files},
        "meta": meta,
    }
