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


def test_direct_convolution_entry_points_check_their_arguments():
    """host-side checks of the direct convolution entry points (no kernel is launched): channel counts outside the built
    variants are DSRG_ERR_INVALID with a message, the weight-gradient workspace is workgroups x cout x 9 x 64 floats (two
    workgroups per CU for 64-channel gradients) and 0 for shapes it does not serve"""
    import ctypes
    from dsrg_amd import _lib
    L = _lib.lib()
    one = ctypes.c_void_p(16)                                             # never dereferenced: the checks come first
    assert L.dsrg_conv3x3_direct_bf16(one, one, None, one, 1, 8, 16, 32, 64, 1, None) != 0
    assert b"channels" in L.dsrg_last_error()
    assert L.dsrg_conv3x3_direct_bf16(None, one, None, one, 1, 8, 16, 64, 64, 1, None) != 0
    assert L.dsrg_conv3x3_wgrad_bf16(one, one, one, one, 1 << 30, 1, 8, 16, 128, 64, None) != 0      # 128 -> 64 has no weight-gradient kernel
    assert L.dsrg_conv3x3_wgrad_workspace(1, 8, 16, 128, 64) == 0 and L.dsrg_conv3x3_wgrad_workspace(1, 8, 16, 64, 32) == 0
    assert L.dsrg_conv3x3_wgrad_workspace(1, 8, 16, 64, 64) == 1 * 64 * 9 * 64 * 4               # one tile -> one workgroup
    assert L.dsrg_conv3x3_wgrad_workspace(1, 8, 16, 128, 128) == 2 * 128 * 9 * 64 * 4            # two 64-channel slices of x
    assert L.dsrg_conv3x3_wgrad_workspace(1, 8, 16, 3, 64) == 64 * 64 * 4
    big = L.dsrg_conv3x3_wgrad_workspace(16, 321, 321, 64, 64)
    assert big % (64 * 9 * 64 * 4) == 0 and big // (64 * 9 * 64 * 4) >= 2                        # capped by the device, not by the 13 776 tiles
    assert L.dsrg_conv3x3_wgrad_bf16(one, one, one, one, 16, 1, 8, 16, 64, 64, None) != 0       # workspace too small
    assert b"workspace" in L.dsrg_last_error()
    assert L.dsrg_maxpool3x3_bwd_relu_bf16(one, one, one, one, one, one, 512, 1, 8, 8, 4, 4, 24, None) != 0   # 24 / 8 does not divide 256


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


def test_annotation_layer_vs_reference_fixture(tmp_path):
    """tests/golden/annotation_cases.npz: the reference's own AnnotationLayer (pylayers.py:346-387: setup, reshape,
    forward, np.random flip) run by tests/golden/make_golden.py on a synthetic cue dictionary; the drop-in class must
    produce the same three tops from the same pickle, ids, images and seed — without and with mirroring"""
    import pickle
    import pylayers
    g = np.load(os.path.join(ROOT, "tests", "golden", "annotation_cases.npz"))
    ids = g["ids"]
    data = {}
    for i in ids.astype(int):
        data['%i_labels' % i], data['%i_cues' % i] = g['%i_labels' % i], g['%i_cues' % i]
    path = os.path.join(str(tmp_path), "cues.pickle")
    pickle.dump(data, open(path, "wb"), protocol=2)

    class Blob(object):
        def __init__(self, a=None):
            self.data = a if a is not None else np.zeros((0,), np.float32)

        def reshape(self, *s):
            self.data = np.zeros(s, np.float32)
    for shape in ((len(ids),), (len(ids), 1, 1, 1)):                       # the id blob as Caffe may shape it
        for tag, mirror in (("plain", False), ("mirror", True)):
            lay = pylayers.AnnotationLayer()
            lay.param_str = "{'cues': %r, 'mirror': %s}" % (path, mirror)
            bottoms = [Blob(ids.copy().reshape(shape)), Blob(g["images"].copy())]
            tops = [Blob(), Blob(), Blob()]
            lay.setup(bottoms, tops)
            lay.reshape(bottoms, tops)
            np.random.seed(int(g["seed"]))
            lay.forward(bottoms, tops)
            assert np.array_equal(tops[0].data, g[tag + "_labels"].astype(np.float32))
            assert np.array_equal(tops[1].data, g[tag + "_cues"].astype(np.float32))
            assert np.array_equal(tops[2].data, g[tag + "_images"])
    with pytest.raises(Exception):
        pylayers.AnnotationLayer().setup([Blob()], [Blob()])               # "The layer needs two inputs!"


def test_more_than_96_labels_is_rejected_at_reshape():
    """beyond kMaxLabels = 96 (the 81-class blobs of the COCO variant fit): a clear Exception from reshape(), not an error
    code from a launch"""
    import pylayers

    class Blob(object):
        def __init__(self, shape):
            self.data = np.zeros(shape, np.float32)

        def reshape(self, *s):
            self.data = np.zeros(s, np.float32)
    for cls, bottoms in [(pylayers.SoftmaxLayer, [Blob((1, 97, 41, 41))]),
                         (pylayers.CRFLayer, [Blob((1, 97, 41, 41)), Blob((1, 3, 321, 321))]),
                         (pylayers.DSRGLayer, [Blob((1, 1, 1, 97)), Blob((1, 97, 41, 41)), Blob((1, 97, 41, 41)), Blob((1, 3, 321, 321))])]:
        lay = cls()
        lay.param_str = "{'th1': 0.99, 'th2': 0.85}"
        lay.setup(bottoms, [Blob((1,))])
        with pytest.raises(Exception, match="at most 96"):
            lay.reshape(bottoms, [Blob((1,))])


def test_annotation_layer_coco(tmp_path):
    """AnnotationLayerCOCO (pylayers.py:387-507): list of image/label pairs -> image-level labels (B,1,1,81), one-hot
    cue planes (B,81,h,w), resized mean-subtracted RGB image; epoch wrap with reshuffle.  Checked against the reference's
    per-pixel loop restated literally."""
    from PIL import Image
    from scipy.ndimage import zoom
    import pylayers
    rng = np.random.default_rng(1)
    root = str(tmp_path) + "/"
    new_h = new_w = 64
    lh = lw = new_h // 8 + 1
    names = []
    for k in range(3):
        img = rng.integers(0, 256, size=(40 + k, 50, 3), dtype=np.uint8)
        lab = rng.integers(0, 81, size=(lh, lw)).astype(np.uint8)
        lab[rng.random((lh, lw)) < 0.2] = 255
        Image.fromarray(img).save(root + "im%d.png" % k)
        Image.fromarray(lab).save(root + "lb%d.png" % k)
        names.append(("im%d.png" % k, "lb%d.png" % k, img, lab))
    with open(root + "list.txt", "w") as f:
        for n in names:
            f.write("%s %s\n" % (n[0], n[1]))
    mean = (104.0, 117.0, 123.0)

    class Blob(object):
        def __init__(self):
            self.data = np.zeros((0,), np.float32)

        def reshape(self, *s):
            self.data = np.zeros(s, np.float32)
    lay = pylayers.AnnotationLayerCOCO()
    lay.param_str = repr({'source': root + "list.txt", 'root': root, 'batch_size': 2, 'mean': mean,
                          'new_size': (new_h, new_w), 'mirror': False})
    tops = [Blob(), Blob(), Blob()]
    lay.setup([], tops)
    assert tops[0].data.shape == (2, 1, 1, 81) and tops[1].data.shape == (2, 81, lh, lw) and tops[2].data.shape == (2, 3, new_h, new_w)
    lay.forward([], tops)
    for n in range(2):
        _, _, img, lab = names[n]
        want_im = zoom(img.astype('float32'), (new_h / float(img.shape[0]), new_w / float(img.shape[1]), 1.0), order=1)
        want_im = (want_im - mean).transpose(2, 0, 1)               # the file is RGB; BGR read then [2,1,0] gives RGB again
        assert np.allclose(tops[2].data[n], want_im, atol=1e-4)
        cues = np.zeros((81, lh, lw), np.uint8)
        for (x, y), v in np.ndenumerate(lab):                        # pylayers.py:489-492
            if not v == 255:
                cues[v, x, y] = 1
        assert np.array_equal(tops[1].data[n], cues)
        want_lab = np.zeros(81)
        want_lab[np.unique(lab[lab != 255])] = 1
        assert np.array_equal(tops[0].data[n, 0, 0], want_lab)
    lay.forward([], tops)                                            # third pair, then the epoch wraps and reshuffles
    assert lay._cur == 1
    lay2 = pylayers.AnnotationLayerCOCO()
    lay2.param_str = lay.param_str.replace("'mirror': False", "'mirror': True")
    lay2.setup([], tops)
    np.random.seed(3)
    im, cues, _ = lay2.load_next_image()
    _, _, img, lab = names[0]
    plain = np.zeros((81, lh, lw), np.uint8)
    ys, xs = np.nonzero(lab != 255)
    plain[lab[ys, xs], ys, xs] = 1
    assert np.array_equal(cues, plain) or np.array_equal(cues, plain[:, :, ::-1])


def test_pylayers_exports_every_reference_class():
    """every layer class of pylayers/pylayers/pylayers.py is importable from the drop-in package"""
    import pylayers
    for name in ("SoftmaxLayer", "CRFLayer", "SeedLossLayer", "BalancedSeedLossLayer", "ConstrainLossLayer",
                 "ExpandLossLayer", "DSRGLayer", "AnnotationLayer", "AnnotationLayerCOCO"):
        cls = getattr(pylayers, name)
        for m in ("setup", "reshape", "forward", "backward"):
            assert callable(getattr(cls, m))
