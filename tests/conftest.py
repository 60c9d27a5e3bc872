import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def golden_srg():
    return np.load(os.path.join(GOLDEN, "srg_cases.npz"))


@pytest.fixture(scope="session")
def golden_glue():
    return np.load(os.path.join(GOLDEN, "layer_glue.npz"))


@pytest.fixture(scope="session")
def golden_cc():
    return np.load(os.path.join(GOLDEN, "cc_cases.npz"))


def srg_case(g, name):
    """-> labels[C] f32, seed[C,H,W] f32, refined[C,H,W] f64, expected[C,H,W] u8"""
    key = name + "_refined_f16"
    refined = g[key].astype(np.float64) if key in g.files else g[name + "_refined"]
    return (g[name + "_labels"].astype(np.float32), g[name + "_seed"].astype(np.float32), refined,
            g[name + "_out"])


def glue_inputs(g, tag):
    """probs with the clip UNDONE (floor values pushed below 1e-4) and mean-subtracted images."""
    from dsrg_amd.synthetic import MEAN_PIXEL
    probs = g[tag + "_probs_clipped"].copy()
    probs[probs == np.float32(1e-4)] = np.float32(5e-5)
    images = g[tag + "_images_u8"].astype(np.float32) - MEAN_PIXEL[None, :, None, None]
    return probs, images
