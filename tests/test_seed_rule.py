"""CPU: the rule the GPU seed comparisons use (conftest.seeds_match_or_borderline) — bit-exact unless a differing pixel
traces to a threshold decision that is borderline on the oracle side."""
import numpy as np
import pytest

from conftest import seeds_match_or_borderline
from dsrg_amd import synthetic as S


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def test_borderline_rule_rejects_real_flips(O):
    """the helper behind the seed comparisons: identical marginals -> bit-exact required; a marginal pushed across a
    threshold by more than eps is NOT excused; one pushed across from within eps is"""
    rng = np.random.default_rng(3)
    labels, cues = S.make_labels_cues(rng, 2, 21, 41, 41)
    probs = O.softmax_forward(S.make_logits(rng, 2, 21, 41, 41))
    refined, _ = O.crf_refine_batch(probs, S.make_images(rng, 2), 12.0, 10)
    want = O.srg_grow_batch(labels, cues, refined)
    assert seeds_match_or_borderline(O, want, labels, cues, refined, refined)[0] == 0
    # find a pixel whose flip changes the seeds: push a confident pixel of a grown region far below th2
    grown = (want - cues).sum(1) > 0
    b, y, x = [int(v[0]) for v in np.where(grown)]
    hip = refined.copy()
    hip[b, :, y, x] = 1.0 / 21
    got = O.srg_grow_batch(labels, cues, hip)
    assert (got != want).any()
    with pytest.raises(AssertionError):
        seeds_match_or_borderline(O, got, labels, cues, refined, hip)
    # the same decision made borderline on the oracle side is excused (and only then)
    cls = np.where(labels[b, 0, 0] == 1)[0]
    k = cls[refined[b, cls, y, x].argmax()]
    ref2 = refined.copy()
    ref2[b, :, y, x] = (1 - (0.85 + 2e-6)) / 20
    ref2[b, k, y, x] = 0.85 + 2e-6
    hip2 = ref2.copy()
    hip2[b, k, y, x] = 0.85 - 2e-6
    got2 = O.srg_grow_batch(labels, cues, hip2)
    nflip, nborder, seeds2 = seeds_match_or_borderline(O, got2, labels, cues, ref2, hip2)
    assert nborder >= 1 and np.array_equal(seeds2, got2)
