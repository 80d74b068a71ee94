"""A seeded sample of tools/soak.py: conversions drawn from everything the
upstream planner accepts, with random sample type, layout, arithmetic, stream
count and step pattern, each stream against the oracle (exact arithmetic bit
for bit, fast arithmetic within its bound)."""
import os
import sys

import numpy as np
import pytest

import support

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
drbamd = pytest.importorskip("drbamd")

sys.path.insert(0, os.path.join(support.ROOT, "tools"))

import soak  # noqa: E402


@pytest.mark.parametrize("seed,arithmetic", [(11, "exact"), (12, "fast"), (13, "both")])
def test_random_conversions_against_the_oracle(seed, arithmetic):
    rng = np.random.default_rng(seed)
    tally = {}
    for _ in range(60):
        case, plan = soak.draw(rng, arithmetic)
        outcome = soak.run_one(case, plan, rng)
        assert outcome in ("fast", "exact", "unsupported"), (case, outcome)
        tally[outcome] = tally.get(outcome, 0) + 1
    assert tally.get("unsupported", 0) < 10, tally


def test_random_conversions_through_the_drop_in():
    """The four-function API on the GPU: samples bit for bit and the callbacks'
    frame counts, against the oracle."""
    rng = np.random.default_rng(21)
    tally = {}
    for _ in range(150):
        case, plan = soak.draw(rng, "exact")
        outcome = soak.run_api_case(case, plan, rng)
        assert outcome in ("api", "unsupported"), (case, outcome)
        tally[outcome] = tally.get(outcome, 0) + 1
    assert tally.get("api", 0) > 100, tally
