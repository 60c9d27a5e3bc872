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


def _is_cuda_tensor(x):
    return type(x).__module__.startswith("torch") and getattr(x, "is_cuda", False)


class DenseCRF(object):
    def __init__(self, W, H, nlabels):
        _lib.require_gpu()
        self._h = None
        h = ctypes.c_void_p()
        check(_lib.lib().dsrg_crf_create(int(W), int(H), int(nlabels), ctypes.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                _lib.lib().dsrg_crf_destroy(self._h)
            except Exception:                # interpreter teardown: the module globals may already be gone
                pass
            self._h = None

    def npixels(self):
        return _lib.lib().dsrg_crf_npixels(self._h)

    def nlabels(self):
        return _lib.lib().dsrg_crf_nlabels(self._h)

    # numpy in / numpy out as the Cython class; CUDA tensors in (and `out=` CUDA tensors) stay on the device.

    def set_unary_energy(self, unary_costs):
        if _is_cuda_tensor(unary_costs):
            import torch
            u = unary_costs.reshape(-1).to(torch.float32).contiguous()
            ptr, n = ctypes.c_void_p(u.data_ptr()), u.numel()
        else:
            u = np.ascontiguousarray(unary_costs, dtype=np.float32).ravel()
            ptr, n = u.ctypes.data_as(ctypes.c_void_p), u.size
        if n != self.npixels() * self.nlabels():
            raise ValueError("unary_costs must hold npixels*nlabels floats")
        check(_lib.lib().dsrg_crf_set_unary_energy(self._h, ptr))

    def add_pairwise_energy(self, w1, theta_alpha_1, theta_alpha_2, theta_betta_1, theta_betta_2, theta_betta_3,
                            w2, theta_gamma_1, theta_gamma_2, im):
        if _is_cuda_tensor(im):
            import torch
            im = im.reshape(-1).to(torch.uint8).contiguous()
            ptr, n = ctypes.c_void_p(im.data_ptr()), im.numel()
        else:
            im = np.ascontiguousarray(im, dtype=np.uint8).ravel()
            ptr, n = im.ctypes.data_as(ctypes.c_void_p), im.size
        if n != self.npixels() * 3:
            raise ValueError("im must hold npixels*3 bytes")
        check(_lib.lib().dsrg_crf_add_pairwise_energy(self._h, w1, theta_alpha_1, theta_alpha_2, theta_betta_1,
                                                      theta_betta_2, theta_betta_3, w2, theta_gamma_1,
                                                      theta_gamma_2, ptr))

    def inference(self, n_iters=10, out=None):
        if out is not None:
            if not (_is_cuda_tensor(out) and out.is_contiguous() and out.numel() == self.npixels() * self.nlabels()
                    and out.element_size() == 4):
                raise ValueError("out must be a contiguous float32 CUDA tensor of npixels*nlabels values")
            check(_lib.lib().dsrg_crf_inference(self._h, int(n_iters), ctypes.c_void_p(out.data_ptr())))
            return out
        probs = np.empty(self.npixels() * self.nlabels(), dtype=np.float32)
        check(_lib.lib().dsrg_crf_inference(self._h, int(n_iters), probs.ctypes.data_as(ctypes.c_void_p)))
        return probs

    def map(self, n_iters=10, out=None):
        if out is not None:
            if not (_is_cuda_tensor(out) and out.is_contiguous() and out.numel() == self.npixels() and out.element_size() == 4):
                raise ValueError("out must be a contiguous int32 CUDA tensor of npixels values")
            check(_lib.lib().dsrg_crf_map(self._h, int(n_iters), ctypes.c_void_p(out.data_ptr())))
            return out
        labels = np.empty(self.npixels(), dtype=np.int32)
        check(_lib.lib().dsrg_crf_map(self._h, int(n_iters), labels.ctypes.data_as(ctypes.c_void_p)))
        return labels

    def lattice_size(self, k):
        return _lib.lib().dsrg_crf_lattice_size(self._h, int(k))

    def profile_start(self, max_launches=4096):
        """bracket every launch of this object's dominant kernel with HIP events (bench.py)"""
        check(_lib.lib().dsrg_crf_profile_start(self._h, int(max_launches)))

    def profile_stop(self):
        """-> (summed kernel milliseconds, launches); synchronises"""
        ms, n = ctypes.c_double(0.0), ctypes.c_int32(0)
        check(_lib.lib().dsrg_crf_profile_stop(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value


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


def CRF_device(image, unary, maxiter=10, scale_factor=1.0, color_factor=13, want="marginals"):
    """`CRF()` for a device-resident caller: image (H,W,3) uint8 and unary (H,W,M) float32 CUDA tensors in, a CUDA tensor
    out — (H,W,M) float32 marginals, or with want="map" the (H,W) int32 arg-max labels — without a PCIe round trip.
    Same statements as CRF() (CRF.py:19-37)."""
    import torch
    assert image.shape[:2] == unary.shape[:2]
    H, W = image.shape[:2]
    nlables = unary.shape[2]
    if torch.cuda.current_stream(unary.device) != torch.cuda.default_stream(unary.device):
        torch.cuda.current_stream(unary.device).synchronize()     # the object API works on the null stream
    crf = DenseCRF(W, H, nlables)
    crf.set_unary_energy(-unary.to(torch.float32))
    crf.add_pairwise_energy(10, 80 / scale_factor, 80 / scale_factor, color_factor, color_factor, color_factor,
                            3, 3 / scale_factor, 3 / scale_factor, image)
    if want == "map":
        return crf.map(maxiter, out=torch.empty((H, W), dtype=torch.int32, device=unary.device))
    return crf.inference(maxiter, out=torch.empty((H, W, nlables), dtype=torch.float32, device=unary.device))
