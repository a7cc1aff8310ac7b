import importlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(sub=None):
    name = "softgnss-python_amd" + ("." + sub if sub else "")
    return importlib.import_module(name)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def scene_from_json(text):
    synth = pkg("synth")
    d = json.loads(str(text))
    return synth.Scene(d["seed"], d["sats"], d["fs"])


_REC_CACHE = {}


@pytest.fixture(scope="session")
def default_record():
    """Host-generated default-scene record, long enough for the 400 ms tracking golden."""
    synth = pkg("synth")
    g = load_golden("trk_default.npz")
    key = int(g["n_samples"])
    if key not in _REC_CACHE:
        _REC_CACHE[key] = synth.generate(scene_from_json(g["scene"]), key)
    return _REC_CACHE[key]
