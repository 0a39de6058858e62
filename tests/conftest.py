import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


# Plan-deterministic parity runs (round-3 verdict, item 3): convolution shapes take the plans SHIPPED with the package
# (ddmi_amd/tuned/conv_plans_mi355x.txt, picked up by ddmi_amd._lib) and the heuristic plan where that file has none - never a
# timing-based pick of the box the test happens to run on - so the bf16 rounding order, and the metrics behind every bf16
# threshold, are the same on every box.  A test that wants the autotuner sets DDMI_TUNE_STRICT=0 itself (before the library loads).
os.environ.setdefault("DDMI_TUNE_STRICT", "1")
# The suite's batches are small (1 .. 16), the benchmark's is 512: with the small-batch regime on (ddmi_amd/csrc/net.h: one-launch
# GroupNorm + split-K finish, five-launch AttentionBlock, no dual-source skip convolution at max_batch <= 16) the suite would stop running
# the kernels the headline number is made of.  It stays OFF here; tests/test_gpu_round6.py switches it on for the networks it pins in
# that regime (ddmi_debug_small_batch).
os.environ.setdefault("DDMI_SMALL_BATCH", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    class G:
        def npz(self, name):
            return dict(np.load(os.path.join(GOLDEN, name)))

        def json(self, name):
            with open(os.path.join(GOLDEN, name)) as f:
                return json.load(f)
    return G()


def rel_err(a, b):
    """max |a-b| / max |b|  -- the 'relative fp32 tolerance' used by every parity test."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
