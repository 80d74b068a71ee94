"""Loaders for the committed golden fixtures (tests/golden/)."""
import gzip
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

_streams = None


def streams():
    """Returns (meta dict, npz handle)."""
    global _streams
    if _streams is None:
        data = np.load(os.path.join(GOLDEN, "streams.npz"), allow_pickle=False)
        meta = json.loads(bytes(data["meta"]).decode())
        _streams = (meta, data)
    return _streams


def planner_rows():
    with gzip.open(os.path.join(GOLDEN, "planner.json.gz"), "rt") as handle:
        return json.load(handle)


def example_text(name):
    with open(os.path.join(GOLDEN, name + ".txt")) as handle:
        return handle.read()
