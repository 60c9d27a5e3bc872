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
