"""The fractional stage of the exact kernel takes its output frames two at a time, a whole
number of the rate ratio's periods apart, and reads each coefficient once for both where
they pick the same bank row.  A pair whose rows differ (the fixed-point phase drifts: about
one pair in a million) computes its frames one after the other.  That way is forced for
every pair here (DRB_AMD_SINC_SHARE=2, read when the library first launches the kernel:
hence a process of its own), and the golden streams and the long steps must come out bit
for bit as they do on the usual way."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pairs_of_frames_whose_bank_rows_differ_are_bit_exact_too():
    environment = dict(os.environ, DRB_AMD_SINC_SHARE="2")
    done = subprocess.run(
        [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
         "tests/test_gpu_parity.py::test_golden_streams_bit_exact", "tests/test_gpu_scale.py"],
        cwd=ROOT, env=environment, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert done.returncode == 0, done.stdout[-4000:]
    assert " passed" in done.stdout, done.stdout[-2000:]
