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


MAX_BATCH = 8            # images per batched object (include/dsrg_hip.h: dsrg_crf_create_batch)


class DenseCRF(object):
    def __init__(self, W, H, nlabels, nimages=None):
        """nimages: None = the reference's one-image object; n >= 1 = a batched object (dsrg_crf_create_batch) whose unary,
        image and result buffers hold n same-sized images back to back and whose every launch carries all of them"""
        _lib.require_gpu()
        self._h = None
        h = ctypes.c_void_p()
        if nimages is None:
            check(_lib.lib().dsrg_crf_create(int(W), int(H), int(nlabels), ctypes.byref(h)))
        else:
            check(_lib.lib().dsrg_crf_create_batch(int(W), int(H), int(nlabels), int(nimages), ctypes.byref(h)))
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

    def set_stream(self, stream, asynchronous=False):
        """run this object's copies and kernels on `stream` (a torch.cuda.Stream, a raw hipStream_t or None = the null
        stream); asynchronous: the calls enqueue and return (device tensors only), results are final after synchronize()"""
        raw = getattr(stream, "cuda_stream", stream) or 0
        check(_lib.lib().dsrg_crf_set_stream(self._h, ctypes.c_void_p(raw), int(bool(asynchronous))))

    def synchronize(self):
        check(_lib.lib().dsrg_crf_synchronize(self._h))

    def profile_start(self, max_launches=4096):
        """bracket every launch of this object's dominant kernel with HIP events (bench.py)"""
        check(_lib.lib().dsrg_crf_profile_start(self._h, int(max_launches)))

    def profile_stop(self):
        """-> (summed kernel milliseconds, launches); synchronises"""
        ms, n = ctypes.c_double(0.0), ctypes.c_int32(0)
        check(_lib.lib().dsrg_crf_profile_stop(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value


# the parameter contract of the reference's CRF() (CRF.py:27-35): Potts weights and kernel widths, divided by scale_factor
_BILATERAL_W, _BILATERAL_XY, _GAUSS_W, _GAUSS_XY = 10, 80.0, 3, 3.0


def _crf_object(width, height, labels, neg_unary, image, scale_factor, color_factor):
    crf = DenseCRF(width, height, labels)
    crf.set_unary_energy(neg_unary)
    sxy_b, sxy_g = _BILATERAL_XY / scale_factor, _GAUSS_XY / scale_factor
    crf.add_pairwise_energy(_BILATERAL_W, sxy_b, sxy_b, color_factor, color_factor, color_factor, _GAUSS_W, sxy_g, sxy_g, image)
    return crf


def CRF(image, unary, maxiter=10, scale_factor=1.0, color_factor=13):
    """Mean-field inference in a fully connected CRF with Gaussian edge potentials — the reference's `krahenbuhl2013.CRF`
    (CRF.py:4-37): image (H,W,3) with values in [0,256), unary (H,W,M) scores whose NEGATIVE is the unary energy ->
    (H,W,M) float32 marginals after `maxiter` iterations; bilateral kernel 10 * k(80/s px, color_factor), spatial kernel
    3 * k(3/s px)."""
    assert(image.shape[:2] == unary.shape[:2])
    height, width, labels = unary.shape
    crf = _crf_object(width, height, labels, -unary.ravel().astype('float32'), image.ravel().astype('ubyte'), scale_factor,
                      color_factor)
    return crf.inference(maxiter).reshape((height, width, labels))


def CRF_device(image, unary, maxiter=10, scale_factor=1.0, color_factor=13, want="marginals"):
    """`CRF()` for a device-resident caller: image (H,W,3) uint8 and unary (H,W,M) float32 CUDA tensors in, a CUDA tensor
    out — (H,W,M) float32 marginals, or with want="map" the (H,W) int32 arg-max labels — without a PCIe round trip.
    Same parameters as CRF() (CRF.py:19-37)."""
    import torch
    assert image.shape[:2] == unary.shape[:2]
    H, W, labels = unary.shape
    if torch.cuda.current_stream(unary.device) != torch.cuda.default_stream(unary.device):
        torch.cuda.current_stream(unary.device).synchronize()     # the object API works on the null stream
    crf = _crf_object(W, H, labels, -unary.to(torch.float32), image, scale_factor, color_factor)
    if want == "map":
        return crf.map(maxiter, out=torch.empty((H, W), dtype=torch.int32, device=unary.device))
    return crf.inference(maxiter, out=torch.empty((H, W, labels), dtype=torch.float32, device=unary.device))


def CRF_device_batch(images, unary, maxiter=10, scale_factor=1.0, color_factor=13, want="marginals", crf=None):
    """`CRF_device` for B same-sized images in ONE set of launches (a batched object, at most MAX_BATCH images per object; more
    are taken in chunks): images (B,H,W,3) uint8 and unary (B,H,W,M) float32 CUDA tensors -> (B,H,W,M) float32 marginals or,
    want="map", (B,H,W) int32 labels.  Each image's result equals CRF_device's on that image bit for bit.  crf: a DenseCRF(W, H,
    M, nimages=B) to reuse (its stream setting is kept); otherwise objects come from the library's cache on the null stream."""
    import torch
    B, H, W, M = unary.shape
    assert tuple(images.shape) == (B, H, W, 3)
    # objects made here run on the CALLER's stream (bound for the call, unbound before they go back to the library's cache): no
    # host synchronisation of the caller's stream up front and no detour over the null stream, which serialises every other stream
    cur = torch.cuda.current_stream(unary.device)
    bind = cur if (crf is None and cur != torch.cuda.default_stream(unary.device)) else None
    sxy_b, sxy_g = _BILATERAL_XY / scale_factor, _GAUSS_XY / scale_factor
    out = torch.empty((B, H, W) if want == "map" else (B, H, W, M), dtype=torch.int32 if want == "map" else torch.float32,
                      device=unary.device)
    for b0 in range(0, B, MAX_BATCH):
        n = min(MAX_BATCH, B - b0)
        obj = crf if (crf is not None and B <= MAX_BATCH) else DenseCRF(W, H, M, nimages=n)
        if bind is not None and obj is not crf:
            obj.set_stream(bind)
        try:
            obj.set_unary_energy((-unary[b0:b0 + n].to(torch.float32)).contiguous())
            obj.add_pairwise_energy(_BILATERAL_W, sxy_b, sxy_b, color_factor, color_factor, color_factor, _GAUSS_W, sxy_g, sxy_g,
                                    images[b0:b0 + n].to(torch.uint8).contiguous())
            if want == "map":
                obj.map(maxiter, out=out[b0:b0 + n])
            else:
                obj.inference(maxiter, out=out[b0:b0 + n])
        finally:
            if bind is not None and obj is not crf:
                obj.set_stream(None)
    return out


def CRF_device_many(pairs, maxiter=10, scale_factor=1.0, color_factor=13, want="map", in_flight=4, batch=1):
    """`CRF_device` over many images with `in_flight` of them overlapping on the GPU — the test-time loop of
    training/tools/test-ms.py:84-111 / generate_train_gt.py:78-106 (10 582 images, one CRF each).  The full-resolution CRF is
    ~170 short dependent launches per image (and one host read-back of the lattice sizes): one image at a time leaves most of
    the chip idle and the host waiting.  Here `in_flight` host threads each own a DenseCRF object per image size and a stream
    (dsrg_crf_set_stream); the library calls release the GIL, so the threads' launch sequences interleave on the host and
    their kernels overlap on the device.  pairs: iterable of (image (H,W,3) uint8, unary (H,W,M) float32) CUDA tensors;
    yields the results in order: (H,W) int32 arg-max labels (want="map") or (H,W,M) float32 marginals.
    batch > 1: consecutive pairs of one shape (up to `batch`, at most MAX_BATCH) additionally share ONE batched object call —
    every launch of the build and of the mean-field loop then carries that many images (the loop is launch-bound: ~6 us
    launches for 10 MB each at one image); results are the same bit for bit."""
    import threading
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    import torch
    local = threading.local()
    device = torch.cuda.current_device()
    caller = torch.cuda.current_stream(device)
    sxy_b, sxy_g = _BILATERAL_XY / scale_factor, _GAUSS_XY / scale_factor

    batch = max(1, min(int(batch), MAX_BATCH))

    def work(group, ready):
        """group: list of (image, unary) of one shape -> list of results"""
        torch.cuda.set_device(device)
        if not hasattr(local, "stream"):
            local.stream, local.objs = torch.cuda.Stream(device=device), {}
        H, W, M = group[0][1].shape
        n = len(group)
        key = (H, W, M, n if batch > 1 else None)
        crf = local.objs.get(key)
        if crf is None:
            crf = local.objs[key] = DenseCRF(W, H, M, nimages=n if batch > 1 else None)
            crf.set_stream(local.stream)
        local.stream.wait_event(ready)                      # the inputs were produced on the caller's stream
        with torch.cuda.stream(local.stream):
            un = torch.stack([u for _, u in group]) if n > 1 else group[0][1]
            im = torch.stack([i for i, _ in group]) if n > 1 else group[0][0]
            crf.set_unary_energy((-un.to(torch.float32)).contiguous())
            crf.add_pairwise_energy(_BILATERAL_W, sxy_b, sxy_b, color_factor, color_factor, color_factor, _GAUSS_W, sxy_g, sxy_g,
                                    im.reshape(-1).to(torch.uint8).contiguous())
            if want == "map":
                out = crf.map(maxiter, out=torch.empty((n, H, W), dtype=torch.int32, device=un.device))
            else:
                out = crf.inference(maxiter, out=torch.empty((n, H, W, M), dtype=torch.float32, device=un.device))
        # allocated on this worker's stream, consumed on the caller's: keep the block out of this stream's pool until the
        # caller's queued work is done with it (the call above returned with the worker's stream drained)
        out.record_stream(caller)
        return [out[k] for k in range(n)]

    def groups():
        run = []
        for image, unary in pairs:
            if run and (len(run) == batch or tuple(unary.shape) != tuple(run[0][1].shape)):
                yield run
                run = []
            run.append((image, unary))
        if run:
            yield run

    pending = deque()
    with ThreadPoolExecutor(max_workers=max(1, int(in_flight))) as ex:
        for group in groups():
            ready = torch.cuda.Event()
            ready.record(caller)
            pending.append(ex.submit(work, group, ready))
            if len(pending) > 2 * in_flight:
                for r in pending.popleft().result():        # (every call of the object API returns with its stream drained)
                    yield r
        while pending:
            for r in pending.popleft().result():
                yield r
