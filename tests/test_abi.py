"""CPU: libdsrg_hip.so loads, exports every symbol include/dsrg_hip.h declares, and its
compute entry points fail loudly (no CPU fallback) when no GPU is present."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "dsrg_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dsrg_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    from dsrg_amd import _lib
    L = _lib.lib()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libdsrg_hip.so does not export %s" % n
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"


def test_no_oracle_in_product():
    """The product packages must not import / reference the oracle."""
    for pkg in ("dsrg_amd", "pylayers", "krahenbuhl2013"):
        for dp, _, files in os.walk(os.path.join(ROOT, pkg)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                    text = open(os.path.join(dp, f), errors="replace").read()
                    assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), (dp, f)
                    assert "liboracle" not in text and "dsrg_oracle" not in text, (dp, f)


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dsrg_amd import _lib
    import krahenbuhl2013
    with pytest.raises(_lib.DsrgError):
        krahenbuhl2013.CRF(np.zeros((4, 4, 3)), np.full((4, 4, 2), 0.5, np.float32))
    import pylayers
    lay = pylayers.SoftmaxLayer()

    class Blob(object):
        def __init__(self, a):
            self.data, self.diff = a, np.zeros_like(a)

        def reshape(self, *s):
            pass
    b, t = Blob(np.zeros((1, 21, 4, 4), np.float32)), Blob(np.zeros((1, 21, 4, 4), np.float32))
    lay.setup([b], [t])
    with pytest.raises(Exception):
        lay.forward([b], [t])


def test_layer_protocol_errors_match_reference():
    """wrong bottom counts raise plain Exception (pylayers.py:27-28,57-58,280-281)."""
    import pylayers
    for cls, n in [(pylayers.SoftmaxLayer, 2), (pylayers.CRFLayer, 1), (pylayers.DSRGLayer, 3),
                   (pylayers.BalancedSeedLossLayer, 1), (pylayers.ConstrainLossLayer, 3)]:
        with pytest.raises(Exception):
            cls().setup([None] * n, [None])


def test_annotation_layer_cue_pickle_format(tmp_path):
    """AnnotationLayer (pylayers.py:346-387): '%i_labels' -> class ids, '%i_cues' -> (c,h,w) index triplets,
    background always on, one flip per image applied to cues and image alike (host-side marshalling)."""
    import pickle
    import pylayers
    rng = np.random.default_rng(0)
    data = {}
    for i in (3, 8):
        data['%i_labels' % i] = np.array(sorted(rng.choice(np.arange(1, 21), size=2, replace=False)))
        k = 30
        data['%i_cues' % i] = np.stack([rng.integers(0, 21, k), rng.integers(0, 41, k), rng.integers(0, 41, k)])
    path = os.path.join(str(tmp_path), "cues.pickle")
    pickle.dump(data, open(path, "wb"), protocol=2)

    class Blob(object):
        def __init__(self, a=None):
            self.data = a if a is not None else np.zeros((0,), np.float32)

        def reshape(self, *s):
            self.data = np.zeros(s, np.float32)
    ids = Blob(np.array([3, 8], np.float32).reshape(2, 1, 1, 1))
    imgs = Blob(rng.standard_normal((2, 3, 321, 321)).astype(np.float32))
    for mirror in (False, True):
        lay = pylayers.AnnotationLayer()
        lay.param_str = "{'cues': %r, 'mirror': %s}" % (path, mirror)
        tops = [Blob(), Blob(), Blob()]
        lay.setup([ids, imgs], tops)
        lay.reshape([ids, imgs], tops)
        np.random.seed(0)
        lay.forward([ids, imgs], tops)
        assert tops[0].data.shape == (2, 1, 1, 21) and tops[1].data.shape == (2, 21, 41, 41)
        for n, i in enumerate((3, 8)):
            want = np.zeros(21); want[0] = 1; want[data['%i_labels' % i]] = 1
            assert np.array_equal(tops[0].data[n, 0, 0], want)
            cues = np.zeros((21, 41, 41), np.float32)
            c = data['%i_cues' % i]
            cues[c[0], c[1], c[2]] = 1
            if mirror:
                same = np.array_equal(tops[1].data[n], cues) and np.array_equal(tops[2].data[n], imgs.data[n])
                flipped = np.array_equal(tops[1].data[n], cues[:, :, ::-1]) and \
                    np.array_equal(tops[2].data[n], imgs.data[n][:, :, ::-1])
                assert same or flipped                       # cues and image flip together
            else:
                assert np.array_equal(tops[1].data[n], cues) and np.array_equal(tops[2].data[n], imgs.data[n])
