import os
import sys

import numpy as np
import pytest

# the GPU tests need MIOpen to answer, not to be fast: take its heuristic pick instead of a ~50 s exhaustive search per
# new convolution shape (bench.py leaves the default, its warm-up absorbs the search)
os.environ.setdefault("MIOPEN_FIND_MODE", "2")

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


def seeds_match_or_borderline(O, got_seeds, labels, cues, refined_oracle, refined_hip, th1=0.99, th2=0.85, eps=1e-5,
                              tag=""):
    """north_star: grown seed masks bit-exact.  The HIP seeds must equal the oracle's SRG of the ORACLE's marginals; a
    pixel may differ only if it traces to a threshold decision that is borderline on the oracle side: the decision value
    v = max over the present classes of refined_oracle lies within `eps` of the threshold that applies (th2; th1 as well
    when the arg-max class is background, pylayers.py:251-257).  Proof by construction: re-run the oracle's SRG on the
    oracle's marginals with ONLY those borderline pixels taken from the HIP marginals — the HIP seeds must equal that
    bit for bit (so with no borderline pixel this is plain equality).  Returns (number of differing seed values, number
    of borderline pixels, the seeds the losses have to be checked against)."""
    got_seeds = np.asarray(got_seeds)
    want = O.srg_grow_batch(labels, cues, refined_oracle)
    nflip = int((got_seeds != want).sum())
    B, C = refined_oracle.shape[:2]
    border = np.zeros((B,) + refined_oracle.shape[2:], dtype=bool)
    for b in range(B):
        cls = np.where(np.asarray(labels).reshape(B, -1)[b] == 1)[0]
        if cls.size == 0:
            continue
        sub = refined_oracle[b, cls]
        k = sub.argmax(0)
        v = sub.max(0)
        border[b] = (np.abs(v - th2) < eps) | ((cls[k] == 0) & (np.abs(v - th1) < eps))
    nborder = int(border.sum())
    print("%sseed values differing from the oracle: %d; oracle-side borderline pixels (|v - th| < %g): %d"
          % (tag + ": " if tag else "", nflip, eps, nborder))
    if nflip == 0:
        return 0, nborder, want
    mixed = np.where(border[:, None], refined_hip, refined_oracle)
    want2 = O.srg_grow_batch(labels, cues, mixed)
    bad = int((got_seeds != want2).sum())
    assert bad == 0, ("%d seed values differ from the oracle and %d of them do not trace to a borderline threshold "
                      "decision (%d borderline pixels)" % (nflip, bad, nborder))
    return nflip, nborder, want2
