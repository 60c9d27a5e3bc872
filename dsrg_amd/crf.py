"""Host-side mirror of the reference's dense-CRF Python API over libdsrg_hip.so.

  DenseCRF  <-> Cython class krahenbuhl2013.wrapper.DenseCRF   (CRF/krahenbuhl2013/wrapper.pyx:20-60)
  CRF()     <-> krahenbuhl2013.CRF                             (CRF/krahenbuhl2013/CRF.py:4-37)

Same names, argument meaning and numpy in/out conventions as the reference, so the
callers (pylayers.py:82,326; training/tools/test-ms.py:106) run unchanged.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import check


class DenseCRF(object):
    def __init__(self, W, H, nlabels):
        _lib.require_gpu()
        self._h = None
        h = ctypes.c_void_p()
        check(_lib.lib().dsrg_crf_create(int(W), int(H), int(nlabels), ctypes.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.lib().dsrg_crf_destroy(self._h)
            self._h = None

    def npixels(self):
        return _lib.lib().dsrg_crf_npixels(self._h)

    def nlabels(self):
        return _lib.lib().dsrg_crf_nlabels(self._h)

    def set_unary_energy(self, unary_costs):
        u = np.ascontiguousarray(unary_costs, dtype=np.float32).ravel()
        if u.size != self.npixels() * self.nlabels():
            raise ValueError("unary_costs must hold npixels*nlabels floats")
        check(_lib.lib().dsrg_crf_set_unary_energy(self._h, u.ctypes.data_as(ctypes.c_void_p)))

    def add_pairwise_energy(self, w1, theta_alpha_1, theta_alpha_2, theta_betta_1, theta_betta_2, theta_betta_3,
                            w2, theta_gamma_1, theta_gamma_2, im):
        im = np.ascontiguousarray(im, dtype=np.uint8).ravel()
        if im.size != self.npixels() * 3:
            raise ValueError("im must hold npixels*3 bytes")
        check(_lib.lib().dsrg_crf_add_pairwise_energy(self._h, w1, theta_alpha_1, theta_alpha_2, theta_betta_1,
                                                      theta_betta_2, theta_betta_3, w2, theta_gamma_1,
                                                      theta_gamma_2, im.ctypes.data_as(ctypes.c_void_p)))

    def inference(self, n_iters=10):
        probs = np.empty(self.npixels() * self.nlabels(), dtype=np.float32)
        check(_lib.lib().dsrg_crf_inference(self._h, int(n_iters), probs.ctypes.data_as(ctypes.c_void_p)))
        return probs

    def map(self, n_iters=10):
        labels = np.empty(self.npixels(), dtype=np.int32)
        check(_lib.lib().dsrg_crf_map(self._h, int(n_iters), labels.ctypes.data_as(ctypes.c_void_p)))
        return labels

    def lattice_size(self, k):
        return _lib.lib().dsrg_crf_lattice_size(self._h, int(k))


def CRF(image, unary, maxiter=10, scale_factor=1.0, color_factor=13):
    """Mean-field inference in a fully connected CRF with Gaussian edge potentials.

    image: (H,W,3) values in [0,256); unary: (H,W,M) — returns (H,W,M) float32 marginals.
    Statement-for-statement the reference function (CRF.py:19-37) over the HIP object."""
    assert(image.shape[:2] == unary.shape[:2])
    H, W = image.shape[:2]
    nlables = unary.shape[2]
    crf = DenseCRF(W, H, nlables)
    crf.set_unary_energy(-unary.ravel().astype('float32'))
    crf.add_pairwise_energy(10, 80 / scale_factor, 80 / scale_factor, color_factor, color_factor, color_factor,
                            3, 3 / scale_factor, 3 / scale_factor, image.ravel().astype('ubyte'))
    prediction = crf.inference(maxiter).reshape((H, W, nlables))
    return prediction
