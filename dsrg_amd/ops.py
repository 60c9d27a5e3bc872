"""Torch-tensor front-end of libdsrg_hip.so: device pointers in, device pointers out.

PyTorch is plumbing here (HBM allocations, streams); all arithmetic happens in the
HIP kernels behind the C ABI (include/dsrg_hip.h).  Every function below names the
reference routine it replaces.
"""
import ctypes
import threading

import torch

from . import _lib
from ._lib import CrfParams, check


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _f32c(t, name):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError("%s must be a contiguous float32 CUDA tensor" % name)
    return t


# ---- deferred bias-gradient reductions (dsrg_defer_reductions / dsrg_flush_reductions) -------------------------------------------
# Inside `with deferred_reductions():` the 5-7 us passes that finish a bias gradient (column sums of a data gradient's partial rows,
# the direct kernels' block partials: fifteen per train-s step) are recorded by the library and run in ONE launch when the block
# ends.  This module's part of the contract: every recorded launch gets partial-row scratch of its own (`_partial_rows`: a fresh
# tensor instead of the cached one) and all such scratch stays alive until the flush is enqueued (`_defer_keep`).
_defer = [False]
_defer_keep = []


def _keep_until_flush(*tensors):
    if _defer[0]:
        _defer_keep.extend(t for t in tensors if t is not None)


@__import__("contextlib").contextmanager
def deferred_reductions(enabled=True):
    """inside: bias gradients returned by this module's launches are NOT final until the block ends (nothing may read them; autograd's
    AccumulateGrad adopting a gradient of a parameter whose .grad is None is no read); on exit one launch finishes them all"""
    if not enabled or _defer[0]:
        yield
        return
    L = _lib.lib()
    check(L.dsrg_defer_reductions(1))
    _defer[0] = True
    try:
        yield
    finally:
        _defer[0] = False
        rc = L.dsrg_flush_reductions(_stream())
        L.dsrg_defer_reductions(0)
        del _defer_keep[:]
        check(rc)


class Context(object):
    """dsrg_ctx_t: device workspace for up to max_batch images of shape (C,H,W)."""

    def __init__(self, max_batch, C, H, W):
        _lib.require_gpu()
        self.max_batch, self.C, self.H, self.W = int(max_batch), int(C), int(H), int(W)
        self.device = torch.cuda.current_device()
        h = ctypes.c_void_p()
        check(_lib.lib().dsrg_ctx_create(self.max_batch, self.C, self.H, self.W, ctypes.byref(h)))
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().dsrg_ctx_destroy(h)
            except Exception:                # interpreter teardown: the module globals may already be gone
                pass
            self._h = None

    def profile_start(self, max_launches=4096):
        """bracket every mean-field filter launch with HIP events on the launch stream"""
        check(_lib.lib().dsrg_ctx_profile_start(self._h, int(max_launches)))

    def profile_stop(self):
        """-> (summed filter-kernel milliseconds, number of launches); synchronises"""
        ms, n = ctypes.c_double(0.0), ctypes.c_int32(0)
        check(_lib.lib().dsrg_ctx_profile_stop(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def lattice_sizes(self, B):
        """(M_gaussian, [M_bilateral per image]) of the lattices built by the last CRF call."""
        mg = ctypes.c_int32(0)
        mb = (ctypes.c_int32 * max(B, 1))()
        check(_lib.lib().dsrg_ctx_lattice_sizes(self._h, B, ctypes.byref(mg), mb, _stream()))
        return mg.value, [mb[i] for i in range(B)]

    def lattice_extras(self, B):
        """(X_gaussian, [X_bilateral per image]): splat entries beyond the first of their vertex (bench.py's LDS model)"""
        xg = ctypes.c_int32(0)
        xb = (ctypes.c_int32 * max(B, 1))()
        check(_lib.lib().dsrg_ctx_lattice_extras(self._h, B, ctypes.byref(xg), xb, _stream()))
        return xg.value, [xb[i] for i in range(B)]

    def filter_plan(self, B):
        """(planes per bilateral workgroup, planes per Gaussian workgroup, workgroups, LDS bytes) of a filter launch over B
        images as the launcher plans it now (bench.py's LDS model)"""
        v = [ctypes.c_int32(0) for _ in range(4)]
        check(_lib.lib().dsrg_ctx_filter_plan(self._h, int(B), *[ctypes.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def lattice_dump(self, kind, b=0):
        """One lattice in the reference's own form (tests): kind 0 = Gaussian, 1 = bilateral lattice of image b.
        -> dict(M, keys (M,d) int16, vid (N,d+1) int32, bary (N,d+1) f32, n1 / n2 (d+1,M) int32 with -1 = none)."""
        import numpy as np
        d, N = (2, self.H * self.W) if kind == 0 else (5, self.H * self.W)
        m = ctypes.c_int32(0)
        L = _lib.lib()
        check(L.dsrg_ctx_lattice_dump(self._h, kind, b, ctypes.byref(m), None, None, None, None, None, _stream()))
        M = m.value
        keys = np.empty((M, d), np.int16)
        vid = np.empty((N, d + 1), np.int32)
        bary = np.empty((N, d + 1), np.float32)
        n1 = np.empty((d + 1, M), np.int32)
        n2 = np.empty((d + 1, M), np.int32)
        check(L.dsrg_ctx_lattice_dump(self._h, kind, b, ctypes.byref(m), keys.ctypes.data, vid.ctypes.data, bary.ctypes.data,
                                      n1.ctypes.data, n2.ctypes.data, _stream()))
        return dict(M=M, keys=keys, vid=vid, bary=bary, n1=n1, n2=n2)

    def lattice_norm(self, kind, b=0):
        """norm = 1/sqrt(K 1 + 1e-20) of one lattice (pairwise.cpp:44,54-57), (N,) float32 numpy (tests)"""
        import numpy as np
        out = np.empty(self.H * self.W, np.float32)
        check(_lib.lib().dsrg_ctx_lattice_norm(self._h, kind, b, out.ctypes.data, _stream()))
        return out

    def filter_once(self, kind, q):
        """one application of the normalised kernel `kind` (0 Gaussian, 1 bilateral) to q (B,C,H,W) with the lattices of
        the last CRF call on this context: DenseKernel::filter (pairwise.cpp:63-80) (tests)"""
        _f32c(q, "q")
        out = torch.empty_like(q)
        check(_lib.lib().dsrg_ctx_filter_once(self._h, kind, q.shape[0], _ptr(q), _ptr(out), _stream()))
        return out

    def read_refined(self, B):
        """float64 marginals (B,C,H,W) the last supervision_step on this context thresholded (tests)"""
        out = torch.empty((B, self.C, self.H, self.W), dtype=torch.float64, device="cuda")
        check(_lib.lib().dsrg_ctx_read_refined(self._h, B, _ptr(out), _stream()))
        return out


_CTX_CACHE = {}
_CTX_LOCK = threading.Lock()


# The LDS-resident path (contexts, the fused step) provisions every lattice for its WORST case — 6 * 4 * ceil(N / 4) vertices for
# N = H * W pixels, whatever the image — so which maps it takes is a function of N alone (include/dsrg_hip.h): N <= 4488, i.e.
# 41x41, 65x65, 66x68; 67x67 is the first square map beyond.  Larger maps take the global-memory path through the same Python
# entry points (crf_refine, supervision_step, dsrg_supervision_loss): the batched full-resolution CRF + the stand-alone
# kernels of the other layers, composed in the order of train-s.prototxt:746-810.
LDS_PATH_MAX_PIXELS = 4488


def lds_path_supports(H, W):
    return H * W <= LDS_PATH_MAX_PIXELS


def get_context(B, C, H, W):
    """the cached workspace of (device, host thread, C, H, W): a context is used by one host thread at a time
    (include/dsrg_hip.h), so every thread gets its own"""
    key = (torch.cuda.current_device(), threading.get_ident(), C, H, W)
    with _CTX_LOCK:
        ctx = _CTX_CACHE.get(key)
        if ctx is None or ctx.max_batch < B:
            ctx = Context(max(B, 1), C, H, W)
            _CTX_CACHE[key] = ctx
    return ctx


def softmax_forward(x):
    """SoftmaxLayer.forward (pylayers.py:46-47)."""
    _f32c(x, "x")
    B, C, H, W = x.shape
    p = torch.empty_like(x)
    check(_lib.lib().dsrg_softmax_forward(B, C, H * W, _ptr(x), _ptr(p), _stream()))
    return p


def softmax_backward(x, top_diff):
    """SoftmaxLayer.backward (pylayers.py:49-51)."""
    _f32c(x, "x"), _f32c(top_diff, "top_diff")
    B, C, H, W = x.shape
    dx = torch.empty_like(x)
    check(_lib.lib().dsrg_softmax_backward(B, C, H * W, _ptr(x), _ptr(top_diff), _ptr(dx), _stream()))
    return dx


def crf_refine(probs, images, scale_factor=12.0, maxiter=10, ctx=None, want_log=True, prepared=False):
    """CRFLayer.forward / DSRGLayer.refinement (pylayers.py:63-88,310-331).

    probs (B,C,H,W) f32 is clipped IN PLACE (as the reference does to its bottom blob);
    returns (refined float64 (B,C,H,W), log-marginals float32 or None).  prepared: the lattices of `images` were built
    by crf_prepare on this context (e.g. on a side stream under the backbone forward)."""
    _f32c(probs, "probs"), _f32c(images, "images")
    B, C, H, W = probs.shape
    if images.shape[0] != B or images.shape[1] != 3:
        raise ValueError("images must be (B,3,Hi,Wi)")
    if not lds_path_supports(H, W):
        return _crf_refine_large(probs, images, scale_factor, maxiter, want_log)
    ctx = ctx or get_context(B, C, H, W)
    refined = torch.empty((B, C, H, W), dtype=torch.float64, device=probs.device)
    logq = torch.empty_like(probs) if want_log else None
    prm = CrfParams.from_crf_args(maxiter, scale_factor)
    check(_lib.lib().dsrg_crf_refine_batch(ctx._h, B, _ptr(probs), None if prepared else _ptr(images), images.shape[2],
                                           images.shape[3], ctypes.byref(prm), _ptr(refined), _ptr(logq), _stream()))
    return refined, logq


_MEAN_PIXEL = (104.0, 117.0, 123.0)


def _map_images_u8(images, H, W):
    """(B,3,Hi,Wi) mean-subtracted float images -> (B,H,W,3) uint8 at the map's size: zoom(order=1) with the (in-1)/(out-1)
    mapping evaluated in double, rounded once to float, + mean pixel, np.round, astype(ubyte) (pylayers.py:70-75) — what
    map_pixel_rgb (csrc/embed.h) does inside the LDS path's embedding kernel"""
    import torch.nn.functional as F
    x = images
    if x.shape[2] != H or x.shape[3] != W:
        x = F.interpolate(x.double(), size=(H, W), mode="bilinear", align_corners=True).float()
    mean = torch.tensor(_MEAN_PIXEL, dtype=torch.float64, device=x.device).view(1, 3, 1, 1)
    v = torch.round(x.double() + mean).to(torch.int64) & 0xff                    # half-even, then C's conversion to unsigned char
    return v.to(torch.uint8).permute(0, 2, 3, 1).contiguous()


def _crf_refine_large(probs, images, scale_factor, maxiter, want_log):
    """crf_refine beyond the LDS path: the in-place clip, the mean field through the batched global-memory objects
    (crf.CRF_device_batch; one image per object when a batch's spatial kernel is too narrow for the image tag), then the
    reference's float64 clip / renormalisation / log (pylayers.py:84-88)"""
    from .crf import CRF_device, CRF_device_batch
    probs.clamp_(min=1e-4)                                                       # pylayers.py:65-67
    B, C, H, W = probs.shape
    im = _map_images_u8(images, H, W)
    un = probs.permute(0, 2, 3, 1).contiguous()
    try:
        q = CRF_device_batch(im, un, maxiter, scale_factor)
    except _lib.DsrgError as e:
        if e.code != _lib.ERR_UNSUPPORTED:                                       # out of memory, a HIP fault: not ours to retry
            raise
        q = torch.stack([CRF_device(im[b], un[b], maxiter, scale_factor) for b in range(B)])
    q64 = q.permute(0, 3, 1, 2).double().clamp_min(1e-4)
    refined = (q64 / q64.sum(1, keepdim=True)).contiguous()
    return refined, (torch.log(refined).float().contiguous() if want_log else None)


def crf_prepare(images, C, H, W, scale_factor=12.0, maxiter=10, ctx=None):
    """Image-dependent half of the CRF (image resampling + bilateral lattice build) on the current
    stream; a later supervision_step(..., prepared=True) on the same context skips it.  Lets a trainer
    hide the lattice build under the backbone forward (side stream).  It overwrites the context's lattices: on a side
    stream, order it behind the last mean field that reads them (`side.wait_stream(main)`, as DSRGTrainer.step does)."""
    _f32c(images, "images")
    B = images.shape[0]
    if not lds_path_supports(H, W):
        return None                      # the global-memory path builds its lattices inside the call that uses them
    ctx = ctx or get_context(B, C, H, W)
    prm = CrfParams.from_crf_args(maxiter, scale_factor)
    check(_lib.lib().dsrg_crf_prepare_batch(ctx._h, B, _ptr(images), images.shape[2], images.shape[3],
                                            ctypes.byref(prm), _stream()))
    return ctx


def crf_meanfield(unary, im_u8, maxiter=10, scale_factor=1.0, color_factor=13, ctx=None):
    """krahenbuhl2013.CRF on device blobs: unary (B,C,H,W) f32 (the `unary` argument of CRF.py:28,
    i.e. minus the energy), im_u8 (B,H,W,3) uint8 -> marginals (B,C,H,W) f32."""
    _f32c(unary, "unary")
    if not (im_u8.is_cuda and im_u8.dtype == torch.uint8 and im_u8.is_contiguous()):
        raise ValueError("im_u8 must be a contiguous uint8 CUDA tensor")
    B, C, H, W = unary.shape
    ctx = ctx or get_context(B, C, H, W)
    q = torch.empty_like(unary)
    prm = CrfParams.from_crf_args(maxiter, scale_factor, color_factor)
    check(_lib.lib().dsrg_crf_meanfield_batch(ctx._h, B, _ptr(unary), _ptr(im_u8), ctypes.byref(prm), _ptr(q),
                                              _stream()))
    return q


def crf_layer_backward(refined, top_diff):
    """CRFLayer.backward (pylayers.py:90-92)."""
    _f32c(top_diff, "top_diff")
    if not (refined.is_cuda and refined.dtype == torch.float64 and refined.is_contiguous()):
        raise ValueError("refined must be a contiguous float64 CUDA tensor")
    out = torch.empty_like(top_diff)
    check(_lib.lib().dsrg_crf_layer_backward(refined.numel(), _ptr(refined), _ptr(top_diff), _ptr(out), _stream()))
    return out


def srg_grow(labels, cues, refined, th1=0.99, th2=0.85):
    """DSRGLayer.generate_seed -> generate_seed_step (pylayers.py:237-275,333-344)."""
    _f32c(labels, "labels"), _f32c(cues, "cues")
    if not (refined.is_cuda and refined.dtype == torch.float64 and refined.is_contiguous()):
        raise ValueError("refined must be a contiguous float64 CUDA tensor")
    B, C, H, W = cues.shape
    if labels.numel() != B * C:
        raise ValueError("labels must hold B*C values")
    seeds = torch.empty_like(cues)
    scratch = torch.empty(B * H * W, dtype=torch.int16, device=cues.device)
    check(_lib.lib().dsrg_srg_grow_batch(B, C, H, W, _ptr(labels), _ptr(cues), _ptr(refined), float(th1), float(th2),
                                         _ptr(seeds), _ptr(scratch), _stream()))
    return seeds


def seed_loss(probs, seeds, want_grad=True, want_loss=True):
    """BalancedSeedLossLayer.forward/backward (pylayers.py:147-152) -> (loss[1] or None, grad or None)."""
    _f32c(probs, "probs"), _f32c(seeds, "seeds")
    B, C, H, W = probs.shape
    loss = torch.empty(1, dtype=torch.float32, device=probs.device) if want_loss else None
    grad = torch.empty_like(probs) if want_grad else None
    check(_lib.lib().dsrg_seed_loss(B, C, H * W, _ptr(probs), _ptr(seeds), _ptr(loss), _ptr(grad), _stream()))
    return loss, grad


def constrain_loss(probs, logq, want_grad=True, want_loss=True):
    """ConstrainLossLayer.forward/backward (pylayers.py:173-180) -> (loss[1] or None, grad_probs, grad_logq)."""
    _f32c(probs, "probs"), _f32c(logq, "logq")
    B, C, H, W = probs.shape
    loss = torch.empty(1, dtype=torch.float32, device=probs.device) if want_loss else None
    gp = torch.empty_like(probs) if want_grad else None
    gq = torch.empty_like(probs) if want_grad else None
    check(_lib.lib().dsrg_constrain_loss(B, C, H * W, _ptr(probs), _ptr(logq), _ptr(loss), _ptr(gp), _ptr(gq),
                                         _stream()))
    return loss, gp, gq


def seed_loss_plain(probs, seeds, want_grad=True):
    """SeedLossLayer forward/backward (pylayers.py:94-118) -> (loss (1,), grad or None)."""
    _f32c(probs, "probs"), _f32c(seeds, "seeds")
    B, C, H, W = probs.shape
    loss = torch.empty(1, dtype=torch.float32, device=probs.device)
    grad = torch.empty_like(probs) if want_grad else None
    check(_lib.lib().dsrg_seed_loss_plain(B, C, H * W, _ptr(probs), _ptr(seeds), _ptr(loss), _ptr(grad), _stream()))
    return loss, grad


def expand_loss(probs, stat, want_grad=True, q_fg=0.996, q_bg=0.999):
    """ExpandLossLayer forward/backward (pylayers.py:183-233): probs (B,C,H,W), stat (B,1,1,C) image-level labels."""
    _f32c(probs, "probs"), _f32c(stat, "stat")
    B, C, H, W = probs.shape
    if stat.numel() != B * C:
        raise ValueError("stat must hold B*C values")
    loss = torch.empty(1, dtype=torch.float32, device=probs.device)
    grad = torch.empty_like(probs) if want_grad else None
    scratch = torch.empty(B * C, dtype=torch.float64, device=probs.device)
    check(_lib.lib().dsrg_expand_loss(B, C, H * W, _ptr(probs), _ptr(stat), float(q_fg), float(q_bg), _ptr(loss), _ptr(grad),
                                      _ptr(scratch), _stream()))
    return loss, grad


def confusion_matrix(gt, pred, nclass, rule_lt=False, hist=None):
    """evaluate.py:25-30 (`add`: gt != 255) / :61-68 (`generateM`: gt < nclass) on uint8 CUDA tensors.
    Returns the (nclass*nclass + 1) int64 counters (added to `hist` when given); the last counts out-of-range labels."""
    if not (gt.is_cuda and pred.is_cuda and gt.dtype == torch.uint8 and pred.dtype == torch.uint8):
        raise ValueError("gt and pred must be uint8 CUDA tensors")
    gt, pred = gt.contiguous(), pred.contiguous()
    if gt.numel() != pred.numel():
        raise ValueError("gt and pred differ in size")
    if hist is None:
        hist = torch.zeros(nclass * nclass + 1, dtype=torch.int64, device=gt.device)
    check(_lib.lib().dsrg_confusion_matrix(gt.numel(), _ptr(gt), _ptr(pred), int(nclass), int(bool(rule_lt)), _ptr(hist),
                                           _stream()))
    return hist


def supervision_step(logits, images, labels, cues, th1=0.99, th2=0.85, scale_factor=12.0, maxiter=10,
                     ctx=None, want_blobs=False, prepared=False):
    """The five Python layers of train-s.prototxt:746-810, forward and backward, in one
    stream-ordered launch sequence (CRF computed once).

    Returns (losses[2] = {loss-Seed, loss-Constrain}, d(sum)/d logits, blobs or None)."""
    _f32c(logits, "logits"), _f32c(images, "images"), _f32c(labels, "labels"), _f32c(cues, "cues")
    B, C, H, W = logits.shape
    if images.dim() != 4 or images.shape[0] != B or images.shape[1] != 3:
        raise ValueError("images must be (B,3,Hi,Wi) with the batch size of logits")
    if labels.numel() != B * C or tuple(cues.shape) != (B, C, H, W):
        raise ValueError("labels must hold B*C values and cues must have the shape of logits")
    if not lds_path_supports(H, W):
        # maps beyond the LDS path: the same five layers, layer by layer, in the prototxt's order and with its gradient wiring
        # (seed-loss gradient + constrain gradient wrt p + CRFLayer.backward(constrain gradient wrt log q), then
        # SoftmaxLayer.backward)
        probs = softmax_forward(logits)
        refined, logq = _crf_refine_large(probs, images, scale_factor, maxiter, True)
        seeds = srg_grow(labels.reshape(B, -1).contiguous(), cues, refined, th1, th2)
        l_seed, g_seed = seed_loss(probs, seeds)
        l_con, g_p, g_q = constrain_loss(probs, logq)
        grad = softmax_backward(logits, g_seed + g_p + crf_layer_backward(refined, g_q))
        losses = torch.stack([l_seed.reshape(()), l_con.reshape(())]).float()
        return losses, grad, (dict(probs=probs, seeds=seeds, logq=logq, refined=refined) if want_blobs else None)
    ctx = ctx or get_context(B, C, H, W)
    losses = torch.empty(2, dtype=torch.float32, device=logits.device)
    grad = torch.empty_like(logits)
    blobs = None
    if want_blobs:
        blobs = dict(probs=torch.empty_like(logits), seeds=torch.empty_like(logits), logq=torch.empty_like(logits))
    prm = CrfParams.from_crf_args(maxiter, scale_factor)
    check(_lib.lib().dsrg_supervision_step(
        ctx._h, B, _ptr(logits), None if prepared else _ptr(images), images.shape[2], images.shape[3],
        _ptr(labels), _ptr(cues),
        float(th1), float(th2), ctypes.byref(prm), _ptr(losses), _ptr(grad),
        _ptr(blobs["probs"]) if blobs else None, _ptr(blobs["seeds"]) if blobs else None,
        _ptr(blobs["logq"]) if blobs else None, _stream()))
    return losses, grad, blobs


def im2col3x3_nhwc(x_nhwc, dilation):
    """(B,H,W,C) contiguous bf16/fp16 (C % 8 == 0) or float32 (C % 4 == 0) -> (B*H*W, 9*C) im2col matrix of a 3x3 'same'
    dilated convolution.  The kernel moves 16-byte channel groups, so a float32 pixel is passed as 2*C two-byte elements."""
    es = x_nhwc.element_size()
    if not (x_nhwc.is_cuda and x_nhwc.is_contiguous() and es in (2, 4)):
        raise ValueError("x must be a contiguous 2- or 4-byte CUDA tensor in NHWC order")
    B, H, W, C = x_nhwc.shape
    out = torch.empty((B * H * W, 9 * C), dtype=x_nhwc.dtype, device=x_nhwc.device)
    check(_lib.lib().dsrg_im2col3x3_nhwc16(_ptr(x_nhwc), _ptr(out), B, H, W, C * (es // 2), int(dilation), _stream()))
    return out


def col2im3x3_nhwc(cols, B, H, W, C, dilation):
    """(B*H*W, 9*C) contiguous bf16 -> (B,C,H,W) channels_last bf16: the adjoint of im2col3x3_nhwc"""
    if not (cols.is_cuda and cols.is_contiguous() and cols.dtype == torch.bfloat16 and cols.shape == (B * H * W, 9 * C)):
        raise ValueError("cols must be a contiguous (B*H*W, 9*C) bf16 CUDA tensor")
    out = torch.empty((B, C, H, W), dtype=cols.dtype, device=cols.device, memory_format=torch.channels_last)
    check(_lib.lib().dsrg_col2im3x3_nhwc_bf16(_ptr(cols), _ptr(out), B, H, W, C, int(dilation), _stream()))
    return out


_PARTIAL_BLOCKS = 512
# scratch for the per-block partial column sums, one buffer per (device, channel count).  Calls are stream-ordered on
# torch's current stream (autograd runs a backward pass on one stream), so consecutive users never overlap; code that
# drives these ops from several streams at once must give each stream its own buffers.
_partials = {}


def _partial_rows(device, C):
    """the partial-row scratch of a bias-gradient launch: the cached buffer of (device, C) — or, while the finishing passes are being
    deferred (deferred_reductions), a buffer of the launch's own, kept until the flush"""
    if _defer[0]:
        part = torch.empty(_PARTIAL_BLOCKS * C, dtype=torch.float32, device=device)
        _defer_keep.append(part)
        return part
    part = _partials.get((device, C))
    if part is None:
        part = _partials[(device, C)] = torch.empty(_PARTIAL_BLOCKS * C, dtype=torch.float32, device=device)
    return part


def _dest(out, shape, dtype, device, channels_last=False):
    """`out` if it can take a kernel's result of this shape / dtype in place (a reducer's gradient slot: dense, the layout the
    kernel writes), else a fresh tensor"""
    if out is not None and out.dtype == dtype and tuple(out.shape) == tuple(shape) and out.device == device and (
            out.is_contiguous(memory_format=torch.channels_last) if channels_last else out.is_contiguous()):
        return out
    if channels_last:
        return torch.empty(shape, dtype=dtype, device=device, memory_format=torch.channels_last)
    return torch.empty(shape, dtype=dtype, device=device)


def relu_bwd_bias(g, y, scale=1.0, gb_out=None):
    """ReLU backward fused with the bias-gradient reduction of the convolution before it.
    g, y: (B,C,H,W) bf16 channels_last (y = the ReLU output, or the output of ReLU + Dropout with scale = 1/(1-p)).
    Returns (scale * g * (y > 0), its sum over B,H,W as f32 (C,))."""
    B, C, H, W = y.shape
    cl = torch.channels_last
    if not (g.is_cuda and g.dtype == torch.bfloat16 and y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=cl)):
        raise ValueError("relu_bwd_bias needs bf16 channels_last CUDA tensors")
    g = g.contiguous(memory_format=cl)
    gm = torch.empty_like(y)
    gb = _dest(gb_out, (C,), torch.float32, y.device)
    part = _partial_rows(y.device, C)
    check(_lib.lib().dsrg_relu_bwd_bias_bf16(_ptr(g), _ptr(y), _ptr(gm), _ptr(gb), _ptr(part), _PARTIAL_BLOCKS,
                                             B * H * W, C, float(scale), _stream()))
    return gm, gb


def bias_grad(g):
    """(B,C,H,W) bf16 channels_last -> per-channel sums over B,H,W as f32 (C,); C <= 256, or a multiple of 8 up to 2048"""
    B, C, H, W = g.shape
    if not (g.is_cuda and g.dtype == torch.bfloat16):
        raise ValueError("bias_grad needs a bf16 CUDA tensor")
    g = g.contiguous(memory_format=torch.channels_last)
    gb = torch.empty(C, dtype=torch.float32, device=g.device)
    part = _partial_rows(g.device, C)
    if C % 8 == 0:                                   # 16-byte lanes, no mask, nothing stored
        check(_lib.lib().dsrg_relu_bwd_bias_bf16(_ptr(g), None, None, _ptr(gb), _ptr(part), _PARTIAL_BLOCKS,
                                                 B * H * W, C, 1.0, _stream()))
    else:
        check(_lib.lib().dsrg_bias_grad_bf16(_ptr(g), _ptr(gb), _ptr(part), _PARTIAL_BLOCKS, B * H * W, C, _stream()))
    return gb


def avgpool3x3_s1(x):
    """3x3 / stride 1 / pad 1 average over padded windows of a (B,C,H,W) bf16 channels_last tensor (also its own backward)"""
    B, C, H, W = x.shape
    x = x.contiguous(memory_format=torch.channels_last)
    out = torch.empty_like(x)
    check(_lib.lib().dsrg_avgpool3x3_s1_bf16(_ptr(x), _ptr(out), B, H, W, C, _stream()))
    return out


def add_relu(a, b):
    """relu(a + b) of two bf16 CUDA tensors of one shape and memory layout in one pass (fp32 sum, one rounding) — the tail of a
    ResNet bottleneck"""
    if not (a.is_cuda and a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.shape == b.shape and a.numel() % 8 == 0):
        raise ValueError("add_relu needs two bf16 CUDA tensors of one shape, 8 | elements")
    if a.stride() != b.stride() or not (a.is_contiguous() or a.is_contiguous(memory_format=torch.channels_last)):
        a, b = a.contiguous(memory_format=torch.channels_last), b.contiguous(memory_format=torch.channels_last)
    y = torch.empty_like(a)
    check(_lib.lib().dsrg_add_relu_bf16(_ptr(a), _ptr(b), _ptr(y), a.numel(), _stream()))
    return y


def relu_mask(g, y, g2=None):
    """(g (+ g2)) where y > 0, else 0: the backward of add_relu (y its output); bf16 CUDA tensors of y's layout"""
    if not (y.is_cuda and y.dtype == torch.bfloat16 and g.dtype == torch.bfloat16 and g.shape == y.shape and y.numel() % 8 == 0):
        raise ValueError("relu_mask needs bf16 CUDA tensors of one shape, 8 | elements")
    if not (y.is_contiguous() or y.is_contiguous(memory_format=torch.channels_last)):
        raise ValueError("relu_mask: y must be dense")
    mf = torch.contiguous_format if y.is_contiguous() else torch.channels_last

    def dense(t):
        return t if t.stride() == y.stride() else t.contiguous(memory_format=mf)
    g = dense(g)
    g2 = dense(g2) if g2 is not None else None
    gm = torch.empty_like(y)
    check(_lib.lib().dsrg_relu_mask_bf16(_ptr(g), _ptr(g2) if g2 is not None else None, _ptr(y), _ptr(gm), y.numel(), _stream()))
    return gm


DIRECT_CONV_CHANNELS = (64, 128)


def conv3x3_direct(x, weight, bias=None, relu=False):
    """3x3 / stride 1 / pad 1 convolution with 64 or 128 channels on either side, or 3 -> 64: x (B,cin,H,W) bf16 channels_last, weight
    (cout,cin,3,3) bf16 (made channels_last here), bias (cout) f32 or None -> (B,cout,H,W) bf16 channels_last; fp32
    accumulation, bias and ReLU fused"""
    B, C, H, W = x.shape
    cl = torch.channels_last
    cout = weight.shape[0]
    if not (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and tuple(weight.shape) == (cout, C, 3, 3)
            and ((C in DIRECT_CONV_CHANNELS and cout in DIRECT_CONV_CHANNELS) or (C, cout) == (3, 64))):
        raise ValueError("conv3x3_direct needs a bf16 CUDA input and a (cout,cin,3,3) bf16 kernel with cin, cout in (64, 128) or 3 -> 64")
    x = x if x.is_contiguous(memory_format=cl) else x.contiguous(memory_format=cl)
    w = weight if weight.is_contiguous(memory_format=cl) else weight.contiguous(memory_format=cl)
    if bias is not None:
        bias = bias.float().contiguous()
    y = torch.empty((B, cout, H, W), dtype=torch.bfloat16, device=x.device, memory_format=cl)
    check(_lib.lib().dsrg_conv3x3_direct_bf16(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), B, H, W, C, cout, int(bool(relu)), _stream()))
    return y


def conv3x3_direct_dgrad(g, weight_t, mask):
    """conv3x3_direct(g, weight_t) — weight_t the flipped, channel-swapped kernel, 64 / 128 channels either side — with the ReLU
    backward of the layer below in the store: mask (B,cout,H,W) bf16 channels_last, that layer's output -> (gx bf16 masked by
    mask > 0, the layer's bias gradient (cout) f32)"""
    B, C, H, W = g.shape
    cl = torch.channels_last
    cout = weight_t.shape[0]
    if not (g.is_cuda and g.dtype == torch.bfloat16 and weight_t.dtype == torch.bfloat16 and tuple(weight_t.shape) == (cout, C, 3, 3)
            and C in DIRECT_CONV_CHANNELS and cout in DIRECT_CONV_CHANNELS and mask.dtype == torch.bfloat16
            and tuple(mask.shape) == (B, cout, H, W) and mask.is_contiguous(memory_format=cl)):
        raise ValueError("conv3x3_direct_dgrad needs bf16 CUDA tensors, 64 / 128 channels, a channels_last mask of the output's shape")
    g = g if g.is_contiguous(memory_format=cl) else g.contiguous(memory_format=cl)
    w = weight_t if weight_t.is_contiguous(memory_format=cl) else weight_t.contiguous(memory_format=cl)
    L = _lib.lib()
    gx = torch.empty((B, cout, H, W), dtype=torch.bfloat16, device=g.device, memory_format=cl)
    gb = torch.empty(cout, dtype=torch.float32, device=g.device)
    ws = torch.empty(L.dsrg_conv3x3_direct_dgrad_workspace(cout), dtype=torch.uint8, device=g.device)
    _keep_until_flush(ws)
    check(L.dsrg_conv3x3_direct_dgrad_bf16(_ptr(g), _ptr(w), _ptr(mask), _ptr(gx), _ptr(gb), _ptr(ws), ws.numel(), B, H, W, C, cout,
                                           _stream()))
    return gx, gb


WGRAD_CONV_SHAPES = ((3, 64), (64, 64), (64, 128), (128, 128))      # (cin, cout)
_wgrad_ws = {}


# ---- the packed bf16 kernels of the float32 master parameters, kept from step to step — only for parameters an optimizer that
# maintains them has claimed (keep_weight_packs: trainer.CaffeSGD).  A convolution node packs the kernel the first time it sees
# such a parameter and leaves the buffers here; the optimizer's update (sgd_pack_step) rewrites them in the pass that updates the
# parameter, so from the second step on a forward finds its packs ready and reads no float32 weight.  An entry is good for exactly
# one value of the parameter: it records the tensor's version counter, which in-place writes bump (load_state_dict, a broadcast)
# — then the node packs again, into the same buffers.  NOT every writer bumps it: torch's private fused optimizer ops
# (torch._fused_sgd_, which torch.optim.SGD(fused=True) calls) leave the counter alone, which is why nothing is kept for a
# parameter unless its optimizer says it plays along; unclaimed parameters are packed inside every forward as before.  Writes
# through `param.data` (whose version counter is its own) or by foreign kernels are equally invisible: code that does that to a
# claimed parameter calls forget_weight_packs afterwards.
import weakref as _weakref


class _WeightPacks(object):
    __slots__ = ("ref", "version", "fwd", "dg", "plain")


_weight_packs = {}                                                    # parameter.data_ptr() -> _WeightPacks


def _packs_entry(weight, plain):
    """the entry of this parameter (whatever value it was packed from), or None"""
    e = _weight_packs.get(weight.data_ptr())
    if e is None:
        return None
    t = e.ref()
    if t is None or t.data_ptr() != weight.data_ptr() or t.shape != weight.shape or e.plain != plain:
        del _weight_packs[weight.data_ptr()]                          # the parameter is gone, another tensor took its address
        return None
    return e


def keep_weight_packs(params, owner):
    """an optimizer's declaration that every write it makes to these parameters either goes through sgd_pack_step or bumps their
    version counter: the convolution nodes may then keep the packed kernels between steps.  The claim lasts as long as `owner`
    (the optimizer object) does — a parameter handed to another optimizer afterwards is packed inside every forward again;
    owner=None withdraws it"""
    ref = _weakref.ref(owner) if owner is not None else None
    for p in params:
        p._dsrg_keep_packs = ref
        _weight_packs.pop(p.data_ptr(), None)     # whatever an earlier owner left may have missed writes made since


def forget_weight_packs(params):
    """drop the kept packed kernels of these parameters (after a write that neither sgd_pack_step made nor a version counter saw)"""
    for p in params:
        _weight_packs.pop(p.data_ptr(), None)


def _packs_claimed(weight):
    ref = getattr(weight, "_dsrg_keep_packs", None)
    return ref is not None and ref() is not None


def _packs_kept(weight):
    # inside a hipGraph capture the packing launches belong in the graph (a replay must see the weights of its own time)
    return _packs_claimed(weight) and weight.is_cuda and not torch.cuda.is_current_stream_capturing()


def _packs_keep(weight, plain, fwd, dg):
    if not _packs_kept(weight):
        return
    if weight.data_ptr() not in _weight_packs:                        # a new parameter (first step of a model): drop those of dead ones
        for k in [k for k, e in _weight_packs.items() if e.ref() is None]:
            del _weight_packs[k]
    e = _WeightPacks()
    e.ref, e.version, e.fwd, e.dg, e.plain = _weakref.ref(weight), weight._version, fwd, dg, plain
    _weight_packs[weight.data_ptr()] = e


def _packs_for(weight, plain, want_fwd, want_dgrad, fwd_shape, dg_shape, fwd_cl=False):
    """-> (fwd, dg, fresh): the buffers of the packed forms wanted; fresh = they already hold this value of the parameter"""
    e = _packs_entry(weight, plain) if _packs_kept(weight) else None
    if e is not None and e.version == weight._version and (e.fwd is not None or not want_fwd) and (e.dg is not None or not want_dgrad):
        return (e.fwd if want_fwd else None), (e.dg if want_dgrad else None), True
    mf = torch.channels_last if fwd_cl else torch.contiguous_format
    fwd = (e.fwd if e is not None and e.fwd is not None else torch.empty(fwd_shape, dtype=torch.bfloat16, device=weight.device, memory_format=mf)) \
        if (want_fwd or (e is not None and e.fwd is not None)) else None
    dg = (e.dg if e is not None and e.dg is not None else torch.empty(dg_shape, dtype=torch.bfloat16, device=weight.device, memory_format=mf)) \
        if (want_dgrad or (e is not None and e.dg is not None)) else None
    return fwd, dg, False


# DSRG_CHECK_PACKS=N (debugging aid): every N-th time a forward takes kept packs for good, the parameter is packed again into
# scratch buffers and compared — a writer the version counter cannot see (`p.data.copy_`, an EMA through `.data`, a foreign
# kernel, torch's private fused optimizer ops) then fails loudly instead of training on kernels one update behind.  0 = off.
_CHECK_PACKS = int(__import__("os").environ.get("DSRG_CHECK_PACKS", "0") or 0)
_check_packs_calls = [0]


def _check_kept_packs(weight, fwd, dg, pack):
    """pack(fwd_buffer_or_None, dg_buffer_or_None) packs `weight` into the buffers given"""
    if _CHECK_PACKS <= 0:
        return
    _check_packs_calls[0] += 1
    if _check_packs_calls[0] % _CHECK_PACKS:
        return
    f2 = torch.empty_like(fwd) if fwd is not None else None
    d2 = torch.empty_like(dg) if dg is not None else None
    pack(f2, d2)
    if (f2 is not None and not torch.equal(f2, fwd)) or (d2 is not None and not torch.equal(d2, dg)):
        raise RuntimeError("DSRG_CHECK_PACKS: the kept bf16 kernels of a %s parameter no longer match its float32 value — it was "
                           "written behind the version counter's back (param.data, a foreign kernel, torch._fused_sgd_); call "
                           "dsrg_amd.ops.forget_weight_packs([param]) after such a write" % (tuple(weight.shape),))


def sgd_pack_step(params, grads, bufs, lrs, wds, momentum):
    """Caffe's SGD update B <- momentum B + (g + wd W); W <- W - lr B of float32 CUDA parameters in one launch per sixteen
    tensors (dsrg_sgd_pack_f32), rewriting the packed bf16 kernels the convolution nodes keep for them in the same pass.
    params / grads / bufs: dense tensors of one memory layout each; lrs / wds: per-tensor rates."""
    import numpy as np
    n = len(params)
    if n == 0:
        return
    ptr = lambda ts: np.array([0 if t is None else t.data_ptr() for t in ts], dtype=np.uint64)
    fwd, dg, shape, entries = [None] * n, [None] * n, np.zeros((n, 4), dtype=np.int32), []
    for i, p in enumerate(params):
        e = _weight_packs.get(p.data_ptr())
        if e is None or e.ref() is None or e.ref().data_ptr() != p.data_ptr() or e.ref().shape != p.shape:
            continue
        fwd[i], dg[i] = e.fwd, e.dg
        shape[i] = (p.shape[0], p.shape[1], p.shape[2] * p.shape[3], e.plain)
        entries.append((e, p))
    pp, gp, bp, fp, dp = ptr(params), ptr(grads), ptr(bufs), ptr(fwd), ptr(dg)
    numel = np.array([p.numel() for p in params], dtype=np.int64)
    lr, wd = np.asarray(lrs, dtype=np.float32), np.asarray(wds, dtype=np.float32)
    a = lambda x: x.ctypes.data
    check(_lib.lib().dsrg_sgd_pack_f32(n, a(pp), a(gp), a(bp), a(fp), a(dp), a(shape), a(numel), a(lr), a(wd), float(momentum), _stream()))
    torch.autograd.graph.increment_version(list(params))              # an in-place write autograd has not seen
    for e, p in entries:
        e.version = p._version


def sgd_pack_eligible(p, g, b):
    """can dsrg_sgd_pack_f32 update this parameter: float32 CUDA tensors of one dense layout"""
    return p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and b.dtype == torch.float32 and g.is_cuda and \
        g.shape == p.shape and all(n == 1 or (sg == sp and sb == sp) for n, sp, sg, sb in zip(p.shape, p.stride(), g.stride(), b.stride())) and \
        (p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))) and \
        ((p.data_ptr() | b.data_ptr()) % 16 == 0 or p.data_ptr() not in _weight_packs)


def pack_direct_weight_pair(weight, want_dgrad=True):
    """the bf16 kernels of conv3x3_direct from a float32 channels_last (cout, cin, 3, 3) parameter with 64 / 128 channels either
    side, one pass: (weight cast, weight flipped with its channel axes swapped — the data gradient's kernel — or None)"""
    cout, cin = weight.shape[0], weight.shape[1]
    cl = torch.channels_last
    if not (weight.is_cuda and weight.dtype == torch.float32 and weight.is_contiguous(memory_format=cl) and tuple(weight.shape[2:]) == (3, 3)
            and cin in DIRECT_CONV_CHANNELS and cout in DIRECT_CONV_CHANNELS):
        raise ValueError("pack_direct_weight_pair needs a float32 channels_last (cout,cin,3,3) CUDA parameter with 64 / 128 channels")
    fwd, dg, fresh = _packs_for(weight, 1, True, want_dgrad, (cout, cin, 3, 3), (cin, cout, 3, 3), fwd_cl=True)
    pack = lambda f, d: check(_lib.lib().dsrg_pack_conv_weight_direct_f32(_ptr(weight.detach()), _ptr(f), _ptr(d), cout, cin, _stream()))
    if not fresh:
        pack(fwd, dg)
        _packs_keep(weight, 1, fwd, dg)
    else:
        _check_kept_packs(weight, fwd, dg, pack)
    return fwd, (dg if want_dgrad else None)


def conv3x3_wgrad(x, g, out_dtype=torch.bfloat16, out=None):
    """weight gradient of the 3x3 / stride 1 / pad 1 convolution y = conv(x, w): x (B,cin,H,W) and g = dL/dy (B,cout,H,W) bf16
    channels_last, (cin, cout) in WGRAD_CONV_SHAPES -> (cout,cin,3,3) bf16 (or float32: out_dtype) channels_last; fp32
    accumulation, deterministic"""
    B, cin, H, W = x.shape
    cout = g.shape[1]
    cl = torch.channels_last
    if not (x.is_cuda and x.dtype == torch.bfloat16 and g.dtype == torch.bfloat16 and tuple(g.shape) == (B, cout, H, W)
            and (cin, cout) in WGRAD_CONV_SHAPES):
        raise ValueError("conv3x3_wgrad needs bf16 CUDA tensors with (cin, cout) in %s" % (WGRAD_CONV_SHAPES,))
    x = x if x.is_contiguous(memory_format=cl) else x.contiguous(memory_format=cl)
    g = g if g.is_contiguous(memory_format=cl) else g.contiguous(memory_format=cl)
    need = _lib.lib().dsrg_conv3x3_wgrad_workspace(B, H, W, cin, cout)
    key = (x.device.index, torch.cuda.current_stream().cuda_stream)
    ws = _wgrad_ws.get(key)                                          # per-stream scratch, reused across layers and steps
    if ws is None or ws.numel() < need:
        ws = _wgrad_ws[key] = torch.empty(need, dtype=torch.uint8, device=x.device)
    if out_dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("conv3x3_wgrad: bf16 or float32 output")
    gw = _dest(out, (cout, cin, 3, 3), out_dtype, x.device, True)
    fn = _lib.lib().dsrg_conv3x3_wgrad_f32 if out_dtype == torch.float32 else _lib.lib().dsrg_conv3x3_wgrad_bf16
    check(fn(_ptr(x), _ptr(g), _ptr(gw), _ptr(ws), ws.numel(), B, H, W, cin, cout, _stream()))
    return gw


def conv_igemm_supported(cin, cout, k):
    """the channel counts / kernel sizes the implicit-GEMM convolution serves (cin % 64 == 0, cout % 256 == 0, k in (1, 3))"""
    return bool(_lib.lib().dsrg_conv_igemm_supported(int(cin), int(cout), int(k)))


def set_igemm_variant(v):
    """tests / tools: the launch form of the implicit-GEMM kernels: 1 = two LDS stages of 64, 2 = ring of four stages of 32, 3 = 1 +
    staggered DMA issue + stream-K where it wins (the default), 4 = stream-K wherever legal, 5 = early barrier, 6 = round 4's launches
    (every K-step multiplied, flat tile maps), 7 = 3 with the weight gradient skipping dead steps of the flat pixel order instead of
    summing over each tap's live pixels; -1 = DSRG_IGEMM_VARIANT or the default.  Same results (6 / 7 / 3: forward bit-identical;
    weight gradient bit-identical between 6 and 7, equal up to fp32 reassociation for 3)"""
    _lib.lib().dsrg_debug_set_igemm_variant(int(v))


def pack_conv_weight(weight, for_dgrad=False, dtype=torch.bfloat16):
    """(cout, cin, k, k) kernel (any float dtype: the fp32 master weights are cast in the same copy) -> the layout
    dsrg_conv_igemm_bf16 reads: (cout, cin / 64, k*k, 64), w_packed[o][cc][tap][c] = w[o][cc*64 + c][tap].  for_dgrad: the kernel
    of the data gradient instead — flipped, channel axes swapped: (cin, cout / 64, k*k, 64)."""
    k = weight.shape[2]
    if for_dgrad:
        weight = weight.flip(2, 3).transpose(0, 1) if k > 1 else weight.transpose(0, 1)
    o, c = weight.shape[0], weight.shape[1]
    out = torch.empty((o, c // 64, k * k, 64), dtype=dtype, device=weight.device)
    if weight.is_contiguous(memory_format=torch.channels_last) and not weight.is_contiguous():
        # the net's parameters are channels_last: memory [o][tap][c] -> [o][c / 64][tap][64] is a strided view of it
        out.copy_(weight.permute(0, 2, 3, 1).reshape(o, k * k, c // 64, 64).permute(0, 2, 1, 3))
    else:
        out.copy_(weight.reshape(o, c // 64, 64, k * k).permute(0, 1, 3, 2))  # cast + permute in one pass
    return out


def pack_conv_weight_pair(weight, want_fwd=True, want_dgrad=True, scale=None):
    """both packed forms of a float32 channels_last (cout, cin, k, k) parameter in one pass (cast included) ->
    (pack_conv_weight(weight), pack_conv_weight(weight, for_dgrad=True)); an entry is None when not wanted.  Other dtypes /
    layouts take the torch copies of pack_conv_weight.  scale (cout) f32: weight[o] * scale[o] is what is packed (the same pass; fresh
    buffers every call — the kept packs of a parameter, which the fused optimizer step rewrites, are those of the bare weight)."""
    cout, cin, k, _ = weight.shape
    if scale is not None:
        if not (weight.is_cuda and weight.dtype == torch.float32 and weight.is_contiguous(memory_format=torch.channels_last)
                and cout % 64 == 0 and cin % 64 == 0 and k in (1, 3)):
            return pack_conv_weight_pair((weight.detach() * scale.view(-1, 1, 1, 1)).contiguous(memory_format=torch.channels_last), want_fwd, want_dgrad)
        fwd = torch.empty((cout, cin // 64, k * k, 64), dtype=torch.bfloat16, device=weight.device) if want_fwd else None
        dg = torch.empty((cin, cout // 64, k * k, 64), dtype=torch.bfloat16, device=weight.device) if want_dgrad else None
        check(_lib.lib().dsrg_pack_conv_weight_scaled_f32(_ptr(weight.detach()), _ptr(_f32c(scale, "scale")), _ptr(fwd), _ptr(dg), cout, cin, k,
                                                          _stream()))
        return fwd, dg
    if not (weight.is_cuda and weight.dtype == torch.float32 and weight.is_contiguous(memory_format=torch.channels_last)
            and cout % 64 == 0 and cin % 64 == 0 and k in (1, 3)):
        return (pack_conv_weight(weight) if want_fwd else None, pack_conv_weight(weight, for_dgrad=True) if want_dgrad else None)
    fwd, dg, fresh = _packs_for(weight, 0, want_fwd, want_dgrad, (cout, cin // 64, k * k, 64), (cin, cout // 64, k * k, 64))
    pack = lambda f, d: check(_lib.lib().dsrg_pack_conv_weight_f32(_ptr(weight.detach()), _ptr(f), _ptr(d), cout, cin, k, _stream()))
    if not fresh:
        pack(fwd, dg)
        _packs_keep(weight, 0, fwd, dg)
    else:
        _check_kept_packs(weight, fwd, dg, pack)
    return (fwd if want_fwd else None), (dg if want_dgrad else None)


def split3_bf16(t, dim):
    """a float32 tensor as three bf16 planes concatenated along `dim`: t = p0 + p1 + p2 up to 2^-24 relative (each plane the bf16
    rounding of what the planes before leave)"""
    p0 = t.to(torch.bfloat16)
    r1 = t - p0.float()
    p1 = r1.to(torch.bfloat16)
    p2 = (r1 - p1.float()).to(torch.bfloat16)
    return torch.cat([p0, p1, p2], dim=dim)


def conv_igemm_split(x, weight, bias, dilation, relu):
    """layer-level prototype: a float32 'same' convolution (3x3 or 1x1, stride 1) on the bf16 MFMA with six bf16 products per
    multiply-add (dsrg_conv_igemm_split_f32): x (B,cin,H,W) float32, weight (cout,cin,k,k) float32, bias (cout) float32 or None ->
    (B,cout,H,W) float32 channels_last.  The operand splits are torch ops here (a production path would have the producing
    epilogue write the planes)."""
    B, cin, H, W = x.shape
    cout, k = weight.shape[0], weight.shape[2]
    x3 = split3_bf16(x.permute(0, 2, 3, 1).contiguous().float(), 3)                     # (B,H,W,3 cin)
    w0, w1, w2 = [pack_conv_weight(p) for p in split3_bf16(weight.float(), 0).split(cout, 0)]
    wv = torch.cat([w0, w1, w0, w2, w0, w1], dim=1).contiguous()                      # (cout, 6 cin / 64, k k, 64)
    y = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    b = bias.float().contiguous() if bias is not None else None
    check(_lib.lib().dsrg_conv_igemm_split_f32(_ptr(x3), _ptr(wv), _ptr(b), _ptr(y), int(dilation), B, H, W, cin, cout, k, int(bool(relu)),
                                               _stream()))
    return y


def conv_igemm(xs, packed, biases, dilations, ksize, relu, dropout_p=0.0, seed=0, stream_k=True):
    """1 .. 4 convolutions of one geometry in one launch (the four ASPP branches): xs[g] (B,cin,H,W) bf16 channels_last,
    packed[g] = pack_conv_weight(w_g), biases[g] (cout) f32 or None -> list of (B,cout,H,W) bf16 channels_last; fp32
    accumulation, bias, ReLU and (dropout_p > 0, multiples of 1/256) the Dropout behind it fused — the mask is a function of
    (seed, branch, position) — no im2col matrix.  stream_k: lend the launch a per-stream scratch so that it may deal its K-steps
    out evenly over the CUs when whole tiles would leave part of the chip idle (same results up to fp32 summation order)"""
    n = len(xs)
    B, cin, H, W = xs[0].shape
    cout = packed[0].shape[0]
    cl = torch.channels_last
    if not (1 <= n <= 4 and len(packed) == n and len(dilations) == n):
        raise ValueError("conv_igemm: 1..4 groups")
    for x, p in zip(xs, packed):
        if not (x.is_cuda and x.dtype == torch.bfloat16 and tuple(x.shape) == (B, cin, H, W) and p.dtype == torch.bfloat16
                and tuple(p.shape) == (cout, cin // 64, ksize * ksize, 64) and p.is_contiguous()):
            raise ValueError("conv_igemm needs bf16 CUDA inputs of one shape and kernels packed by pack_conv_weight")
    xs = [x if x.is_contiguous(memory_format=cl) else x.contiguous(memory_format=cl) for x in xs]
    bs = [None if b is None else _f32c(b, "bias") for b in (biases or [None] * n)]
    ys = [torch.empty((B, cout, H, W), dtype=torch.bfloat16, device=xs[0].device, memory_format=cl) for _ in range(n)]
    vp = ctypes.c_void_p * n
    ws = _igemm_sk_workspace(xs[0].device) if stream_k else None
    check(_lib.lib().dsrg_conv_igemm_bf16(vp(*[x.data_ptr() for x in xs]), vp(*[p.data_ptr() for p in packed]),
                                          vp(*[None if b is None else b.data_ptr() for b in bs]),
                                          vp(*[y.data_ptr() for y in ys]), (ctypes.c_int * n)(*[int(d) for d in dilations]),
                                          n, B, H, W, cin, cout, ksize, int(bool(relu)), float(dropout_p),
                                          int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(ws), ws.numel() if ws is not None else 0, _stream()))
    return ys


def conv_igemm_residual(x, packed, bias, res, mask, dilation, ksize, relu):
    """one convolution with a residual in its store (dsrg_conv_igemm_residual_bf16): x (B,cin,H,W) bf16 channels_last, packed =
    pack_conv_weight(w), bias (cout) f32 or None, res (B,cout,H,W) bf16 channels_last, mask the same or None ->
    post(bf16(conv + bias) + res), post = ReLU (relu) and / or zero where mask <= 0.  Forward of a residual block's last
    convolution (res = the shortcut); data gradient of its first (res = the shortcut's gradient, mask = the block input)."""
    B, cin, H, W = x.shape
    cout = packed.shape[0]
    cl = torch.channels_last
    ok = lambda t: t.is_cuda and t.dtype == torch.bfloat16 and tuple(t.shape) == (B, cout, H, W) and t.is_contiguous(memory_format=cl)   # noqa: E731
    if not (x.is_cuda and x.dtype == torch.bfloat16 and packed.dtype == torch.bfloat16 and packed.is_contiguous()
            and tuple(packed.shape) == (cout, cin // 64, ksize * ksize, 64) and ok(res) and (mask is None or ok(mask))):
        raise ValueError("conv_igemm_residual needs bf16 channels_last tensors of matching shapes and a kernel packed by pack_conv_weight")
    x = x if x.is_contiguous(memory_format=cl) else x.contiguous(memory_format=cl)
    b = None if bias is None else _f32c(bias, "bias")
    y = torch.empty((B, cout, H, W), dtype=torch.bfloat16, device=x.device, memory_format=cl)
    check(_lib.lib().dsrg_conv_igemm_residual_bf16(_ptr(x), _ptr(packed), _ptr(b), _ptr(res), _ptr(mask), _ptr(y), int(dilation), B, H, W,
                                                   cin, cout, ksize, int(bool(relu)), _stream()))
    return y


def aspp_shift_sum(y, offsets, outputs, bias=None):
    """the gather half of an ASPP head run as one 1x1 convolution (dsrg_aspp_shift_sum_f32): y (B,CT,H,W) bf16 channels_last = the
    1x1 product of the feature map with all (branch, tap) kernels stacked (pair j's outputs at channels j * outputs ..), offsets a list
    of (dy, dx) per pair -> (B,outputs,H,W) float32 (channels_last strides): out[.,o,y,x] = bias[o] + sum_j y[., j outputs + o, y + dy_j,
    x + dx_j], zero outside the map"""
    B, CT, H, W = y.shape
    J = len(offsets)
    if not (y.is_cuda and y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last) and 1 <= J <= 36 and J * outputs <= CT):
        raise ValueError("aspp_shift_sum needs a bf16 channels_last (B,CT,H,W) tensor with CT >= pairs * outputs, at most 36 pairs")
    out = torch.empty((B, H, W, outputs), dtype=torch.float32, device=y.device)
    off = (ctypes.c_int * (2 * J))(*[int(v) for pair in offsets for v in pair])
    b = None if bias is None else _f32c(bias, "bias")
    check(_lib.lib().dsrg_aspp_shift_sum_f32(_ptr(y), _ptr(b), _ptr(out), off, J, int(outputs), CT, B, H, W, _stream()))
    return out.permute(0, 3, 1, 2)


def aspp_shift_gather(g, offsets, channels):
    """the backward of aspp_shift_sum (dsrg_aspp_shift_gather_bf16): g (B,outputs,H,W) float32 -> the gradient of y, (B,channels,H,W) bf16
    channels_last: gp[., j outputs + o, y, x] = bf16(g[., o, y - dy_j, x - dx_j]), zero outside the map and in the channels past J outputs"""
    B, O, H, W = g.shape
    J = len(offsets)
    if not (g.is_cuda and 1 <= J <= 36 and J * O <= channels):
        raise ValueError("aspp_shift_gather: at most 36 pairs, channels >= pairs * outputs")
    gn = g.float().permute(0, 2, 3, 1).contiguous()                                    # (B,H,W,O): a view if g has channels_last strides
    gp = torch.empty((B, channels, H, W), dtype=torch.bfloat16, device=g.device, memory_format=torch.channels_last)
    off = (ctypes.c_int * (2 * J))(*[int(v) for pair in offsets for v in pair])
    check(_lib.lib().dsrg_aspp_shift_gather_bf16(_ptr(gn), _ptr(gp), off, J, O, int(channels), B, H, W, _stream()))
    return gp


def conv_igemm_dgrad(gs, packed_t, masks, dilations, ksize, mask_scale=1.0, bias_grad=True, gb_outs=None):
    """the data gradient of 1 .. 4 convolutions whose inputs were ReLU (+ Dropout) outputs, with that layer's backward folded
    in: gs[g] (B,cout_fwd,H,W) bf16 channels_last, packed_t[g] the flipped + transposed packing, masks[g] (B,cin_fwd,H,W) bf16
    the outputs of the layer below -> (list of (B,cin_fwd,H,W) bf16 = conv_T(g) * mask_scale where mask > 0, list of (cin_fwd)
    f32 bias gradients of the layer below or None)"""
    n = len(gs)
    B, cin, H, W = gs[0].shape
    cout = packed_t[0].shape[0]
    cl = torch.channels_last
    if not (1 <= n <= 4 and len(packed_t) == n and len(dilations) == n and len(masks) == n):
        raise ValueError("conv_igemm_dgrad: 1..4 groups")
    for g, p, m in zip(gs, packed_t, masks):
        if not (g.is_cuda and g.dtype == torch.bfloat16 and tuple(g.shape) == (B, cin, H, W) and p.dtype == torch.bfloat16
                and tuple(p.shape) == (cout, cin // 64, ksize * ksize, 64) and p.is_contiguous() and m.dtype == torch.bfloat16
                and tuple(m.shape) == (B, cout, H, W) and m.is_contiguous(memory_format=cl)):
            raise ValueError("conv_igemm_dgrad needs bf16 channels_last gradients / masks of one shape and packed kernels")
    gs = [g if g.is_contiguous(memory_format=cl) else g.contiguous(memory_format=cl) for g in gs]
    outs = [torch.empty((B, cout, H, W), dtype=torch.bfloat16, device=gs[0].device, memory_format=cl) for _ in range(n)]
    vp = ctypes.c_void_p * n
    L = _lib.lib()
    gb = ws = None
    if bias_grad:
        gb = [_dest(gb_outs[i] if gb_outs is not None else None, (cout,), torch.float32, gs[0].device) for i in range(n)]
        ws = torch.empty(L.dsrg_conv_igemm_dgrad_workspace(n, B, H, W, cout), dtype=torch.uint8, device=gs[0].device)
        _keep_until_flush(ws)
    check(L.dsrg_conv_igemm_dgrad_bf16(vp(*[g.data_ptr() for g in gs]), vp(*[p.data_ptr() for p in packed_t]),
                                       vp(*[m.data_ptr() for m in masks]), vp(*[o.data_ptr() for o in outs]),
                                       vp(*[b.data_ptr() for b in gb]) if gb else None,
                                       (ctypes.c_int * n)(*[int(d) for d in dilations]), n, B, H, W, cin, cout, ksize,
                                       float(mask_scale), _ptr(ws), ws.numel() if ws is not None else 0, _stream()))
    return outs, gb


def conv_igemm_backward(g, packed_d, x, dilation, mask=None, mask_scale=1.0, ksize=3, gw_out=None, gb_out=None):
    """the whole backward of one 3x3 convolution (forward geometry cin -> cout) in one launch (dsrg_conv_igemm_backward_bf16): g
    (B,cout,H,W) and x (B,cin,H,W) bf16 channels_last, packed_d = pack_conv_weight(w, for_dgrad=True); mask: the layer's input when
    it is the sole-consumer ReLU output of the layer below (then that layer's ReLU / Dropout backward and bias gradient ride in
    the data gradient, as conv_igemm_dgrad) -> (gx (B,cin,H,W) bf16 channels_last, gw (cout,cin,k,k) float32 channels_last, bias
    gradient of the layer below (cin) f32 or None).  gx and the bias gradient equal conv_igemm(_dgrad)'s bit for bit, gw equals
    conv_igemm_wgrad's up to fp32 reassociation (another pixel split)."""
    B, cout, H, W = g.shape
    cin = x.shape[1]
    cl = torch.channels_last
    if not (g.is_cuda and g.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and tuple(x.shape) == (B, cin, H, W)
            and packed_d.dtype == torch.bfloat16 and tuple(packed_d.shape) == (cin, cout // 64, ksize * ksize, 64)
            and packed_d.is_contiguous() and x.is_contiguous(memory_format=cl)
            and (mask is None or (mask.dtype == torch.bfloat16 and tuple(mask.shape) == (B, cin, H, W) and mask.is_contiguous(memory_format=cl)))):
        raise ValueError("conv_igemm_backward needs bf16 channels_last g / x (/ mask) of one geometry and the data-gradient packing")
    g = g if g.is_contiguous(memory_format=cl) else g.contiguous(memory_format=cl)
    L = _lib.lib()
    need = L.dsrg_conv_igemm_wgrad_workspace(1, B, H, W, cin, cout, ksize)
    if need == 0:
        raise ValueError("conv_igemm_backward: 256 | cin (or cin = 128), 256 | cout required (got %d, %d)" % (cin, cout))
    key = (g.device.index, torch.cuda.current_stream().cuda_stream)
    ws = _igemm_ws.get(key)                                          # per-stream scratch, shared with conv_igemm_wgrad
    if ws is None or ws.numel() < need:
        ws = _igemm_ws[key] = torch.empty(need, dtype=torch.uint8, device=g.device)
    gx = torch.empty((B, cin, H, W), dtype=torch.bfloat16, device=g.device, memory_format=cl)
    gw = _dest(gw_out, (cout, cin, ksize, ksize), torch.float32, g.device, True)
    gb = cws = None
    if mask is not None:
        gb = _dest(gb_out, (cin,), torch.float32, g.device)
        cws = torch.empty(L.dsrg_conv_igemm_dgrad_workspace(1, B, H, W, cin), dtype=torch.uint8, device=g.device)
        _keep_until_flush(cws)
    check(L.dsrg_conv_igemm_backward_bf16(_ptr(g), _ptr(packed_d), _ptr(x), _ptr(mask), _ptr(gx), _ptr(gw), int(dilation), _ptr(gb),
                                          float(mask_scale), _ptr(cws), cws.numel() if cws is not None else 0, _ptr(ws), ws.numel(),
                                          B, H, W, cin, cout, ksize, _stream()))
    return gx, gw, gb


def conv_igemm_backward_residual(g, packed_d, x, dilation, ksize, mask=None, res=None, gw_scale=None, gw_out=None):
    """conv_igemm_backward (3x3 or 1x1) for a convolution inside a residual block (dsrg_conv_igemm_backward_residual_bf16): res
    (B,cin,H,W) bf16 channels_last or None is added to the bf16-rounded data gradient before the mask (the gradient along the
    block's shortcut); gw_scale (cout) f32 or None multiplies the weight gradient per output channel (the constant scale the
    forward's kernel was packed with); no bias gradient -> (gx, gw)"""
    B, cout, H, W = g.shape
    cin = x.shape[1]
    cl = torch.channels_last
    ok = lambda t: t.dtype == torch.bfloat16 and tuple(t.shape) == (B, cin, H, W) and t.is_contiguous(memory_format=cl)      # noqa: E731
    if not (g.is_cuda and g.dtype == torch.bfloat16 and ok(x) and packed_d.dtype == torch.bfloat16 and packed_d.is_contiguous()
            and tuple(packed_d.shape) == (cin, cout // 64, ksize * ksize, 64) and (mask is None or ok(mask)) and (res is None or ok(res))):
        raise ValueError("conv_igemm_backward_residual needs bf16 channels_last g / x (/ mask / res) of one geometry and the data-gradient packing")
    g = g if g.is_contiguous(memory_format=cl) else g.contiguous(memory_format=cl)
    L = _lib.lib()
    need = L.dsrg_conv_igemm_wgrad_workspace(1, B, H, W, cin, cout, ksize)
    if need == 0:
        raise ValueError("conv_igemm_backward_residual: 256 | cin (or cin = 128 with a 3x3 kernel), 256 | cout required (got %d, %d)" % (cin, cout))
    key = (g.device.index, torch.cuda.current_stream().cuda_stream)
    ws = _igemm_ws.get(key)                                          # per-stream scratch, shared with conv_igemm_wgrad
    if ws is None or ws.numel() < need:
        ws = _igemm_ws[key] = torch.empty(need, dtype=torch.uint8, device=g.device)
    gx = torch.empty((B, cin, H, W), dtype=torch.bfloat16, device=g.device, memory_format=cl)
    gw = _dest(gw_out, (cout, cin, ksize, ksize), torch.float32, g.device, True)
    sc = None if gw_scale is None else _f32c(gw_scale, "gw_scale")
    check(L.dsrg_conv_igemm_backward_residual_bf16(_ptr(g), _ptr(packed_d), _ptr(x), _ptr(mask), _ptr(res), _ptr(gx), _ptr(gw), _ptr(sc),
                                                   int(dilation), _ptr(ws), ws.numel(), B, H, W, cin, cout, ksize, _stream()))
    return gx, gw


_igemm_sk_ws = {}


def _igemm_sk_workspace(device):
    """the stream-K scratch of (device, current stream): one accumulator tile per CU + flags, allocated once"""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _igemm_sk_ws.get(key)
    if ws is None:
        ws = _igemm_sk_ws[key] = torch.empty(_lib.lib().dsrg_conv_igemm_workspace(), dtype=torch.uint8, device=device)
    return ws


def conv_igemm_stream_k_status(device=None):
    """tests: 0, or 1 if a workgroup of the last stream-K launch on the current stream gave up waiting for a partial tile"""
    device = device or torch.device("cuda", torch.cuda.current_device())
    ws = _igemm_sk_workspace(device)
    st = ctypes.c_int(0)
    check(_lib.lib().dsrg_conv_igemm_workspace_status(_ptr(ws), _stream(), ctypes.byref(st)))
    return st.value


def dropout_seed():
    """a 63-bit seed for a fused dropout from torch's CPU generator (deterministic under torch.manual_seed, no device sync)"""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


_igemm_ws = {}


def conv_igemm_wgrad_supported(cin, cout, k):
    """the shapes conv_igemm_wgrad is the recommended route for (full 256 x 256 tiles): 256 | cout, 256 | cin (or cin = 128 for a 3x3
    kernel), k in (1, 3)"""
    cin, cout, k = int(cin), int(cout), int(k)
    return k in (1, 3) and ((cin >= 256 and cin % 256 == 0) or (cin == 128 and k == 3)) and cout >= 256 and cout % 256 == 0


def conv_igemm_wgrad_launchable(cin, cout, k):
    """the shapes the launch takes: any multiples of 64 channels (narrow tensors leave part of a tile empty — fine where the layer is
    bandwidth-bound: ResNet res2 / res3, 64 / 128 channels over 42 - 166 thousand pixels)"""
    return _lib.lib().dsrg_conv_igemm_wgrad_workspace(1, 1, 8, 8, int(cin), int(cout), int(k)) > 0


def conv_igemm_wgrad(xs, gs, dilations, ksize, out_dtype=torch.float32, outs=None):
    """weight gradients of 1 .. 4 convolutions of one geometry in one launch: xs[g] (B,cin,H,W) the layer inputs and gs[g]
    (B,cout,H,W) the output gradients, bf16 channels_last -> list of (cout,cin,k,k) channels_last tensors in float32 (the
    master weights' gradient) or bf16; cin % 256 == 0, cout % 256 == 0; fp32 accumulation, deterministic, no im2col matrix"""
    n = len(xs)
    B, cin, H, W = xs[0].shape
    cout = gs[0].shape[1]
    cl = torch.channels_last
    if not (1 <= n <= 4 and len(gs) == n and len(dilations) == n) or out_dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("conv_igemm_wgrad: 1..4 groups, float32 or bfloat16 result")
    for x, g in zip(xs, gs):
        if not (x.is_cuda and x.dtype == torch.bfloat16 and g.dtype == torch.bfloat16 and tuple(x.shape) == (B, cin, H, W)
                and tuple(g.shape) == (B, cout, H, W)):
            raise ValueError("conv_igemm_wgrad needs bf16 CUDA tensors of one geometry")
    xs = [x if x.is_contiguous(memory_format=cl) else x.contiguous(memory_format=cl) for x in xs]
    gs = [g if g.is_contiguous(memory_format=cl) else g.contiguous(memory_format=cl) for g in gs]
    L = _lib.lib()
    need = L.dsrg_conv_igemm_wgrad_workspace(n, B, H, W, cin, cout, ksize)
    if need == 0:
        raise ValueError("conv_igemm_wgrad: 64 | cin, 64 | cout, k in (1, 3) required (got %d, %d, %d)" % (cin, cout, ksize))
    key = (xs[0].device.index, torch.cuda.current_stream().cuda_stream)
    ws = _igemm_ws.get(key)                                          # per-stream scratch, reused across layers and steps
    if ws is None or ws.numel() < need:
        ws = _igemm_ws[key] = torch.empty(need, dtype=torch.uint8, device=xs[0].device)
    gws = [_dest(outs[i] if outs is not None else None, (cout, cin, ksize, ksize), out_dtype, xs[0].device, True) for i in range(n)]
    vp = ctypes.c_void_p * n
    check(L.dsrg_conv_igemm_wgrad_bf16(vp(*[x.data_ptr() for x in xs]), vp(*[g.data_ptr() for g in gs]),
                                       vp(*[w.data_ptr() for w in gws]), (ctypes.c_int * n)(*[int(d) for d in dilations]), n,
                                       _ptr(ws), ws.numel(), B, H, W, cin, cout, ksize, int(out_dtype == torch.bfloat16), _stream()))
    return gws


def heads_forward(xs, weight, bias):
    """fc8-SEC_k + Eltwise SUM in float32: xs = list of <= 4 (B,K,H,W) bf16 channels_last activations, weight (n,O,K) f32,
    bias (n,O) f32 or None -> (B,O,H,W) float32, NCHW-contiguous (what the supervision path reads)."""
    B, K, H, W = xs[0].shape
    n, O = weight.shape[0], weight.shape[1]
    cl = torch.channels_last
    xs = [x if x.is_contiguous(memory_format=cl) else x.contiguous(memory_format=cl) for x in xs]
    if not all(x.is_cuda and x.dtype == torch.bfloat16 and x.shape == xs[0].shape for x in xs) or len(xs) != n:
        raise ValueError("heads_forward needs n bf16 CUDA activations of one shape")
    _f32c(weight, "weight")
    if bias is not None:
        _f32c(bias, "bias")
    out = torch.empty((B, O, H, W), dtype=torch.float32, device=xs[0].device)
    ptrs = (ctypes.c_void_p * 4)(*([x.data_ptr() for x in xs] + [None] * (4 - n)))
    check(_lib.lib().dsrg_heads_forward_bf16(ptrs, n, _ptr(weight), _ptr(bias), _ptr(out), B, H * W, K, O, _stream()))
    return out


def heads_backward(xs, weight, g, need_gx=True, relu_scale=0.0):
    """backward of heads_forward from the float32 score gradient g (B,O,H,W): -> (list of gx_k (B,K,H,W) bf16 channels_last or
    None, gw (n,O,K) float32).  relu_scale > 0: the x_k are ReLU (+ Dropout, scale relu_scale) outputs nobody else reads — the
    gx_k come out masked by x_k > 0 and scaled, and a third result (n,K) float32 is the bias gradient of the layer below"""
    B, K, H, W = xs[0].shape
    n, O = weight.shape[0], weight.shape[1]
    cl = torch.channels_last
    xs = [x if x.is_contiguous(memory_format=cl) else x.contiguous(memory_format=cl) for x in xs]
    _f32c(weight, "weight")
    g = g.contiguous()
    _f32c(g, "g")
    L = _lib.lib()
    M = B * H * W
    gx = torch.empty((n, B, H, W, K), dtype=torch.bfloat16, device=g.device) if need_gx else None
    gw = torch.empty((n, O, K), dtype=torch.float32, device=g.device)
    part = torch.empty(L.dsrg_heads_backward_chunks(M) * n * O * K, dtype=torch.float32, device=g.device)
    ptrs = (ctypes.c_void_p * 4)(*([x.data_ptr() for x in xs] + [None] * (4 - n)))
    if relu_scale > 0.0 and need_gx:
        gb = torch.empty((n, K), dtype=torch.float32, device=g.device)
        ws = torch.empty(L.dsrg_heads_backward_relu_workspace(n, M, K), dtype=torch.uint8, device=g.device)
        _keep_until_flush(ws)
        check(L.dsrg_heads_backward_relu_bf16(ptrs, n, _ptr(weight), _ptr(g), _ptr(gx), M * K * 2, _ptr(gw), _ptr(part),
                                              float(relu_scale), _ptr(gb), _ptr(ws), ws.numel(), B, H * W, K, O, _stream()))
        return [gx[k].permute(0, 3, 1, 2) for k in range(n)], gw, gb
    check(L.dsrg_heads_backward_bf16(ptrs, n, _ptr(weight), _ptr(g), _ptr(gx), M * K * 2, _ptr(gw), _ptr(part), B, H * W, K, O,
                                     _stream()))
    return ([gx[k].permute(0, 3, 1, 2) for k in range(n)] if need_gx else None), gw


def maxpool3x3_out_size(n, stride, ceil_mode):
    """output extent of a 3x3 / pad 1 pooling window walk over n pixels (Caffe's ceil rule, torch's with ceil_mode)"""
    num = n + 2 - 3
    o = (-(-num // stride) if ceil_mode else num // stride) + 1
    if (o - 1) * stride >= n + 1:                       # the last window must start inside the image or its left pad
        o -= 1
    return o


def maxpool3x3_fwd(x, stride, ceil_mode, relu_input=False):
    """x (B,C,H,W) bf16 channels_last -> (pooled (B,C,OH,OW) channels_last, window codes (B,OH,OW,C) uint8).
    relu_input: x is a ReLU's output and maxpool3x3_bwd_relu will be given these codes WITHOUT x — windows whose maximum is
    not positive get a code that names no position (the ReLU mask rides in the codes; the pooled values are the same)"""
    B, C, H, W = x.shape
    OH, OW = maxpool3x3_out_size(H, stride, ceil_mode), maxpool3x3_out_size(W, stride, ceil_mode)
    out = torch.empty((B, C, OH, OW), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    code = torch.empty((B, OH, OW, C), dtype=torch.uint8, device=x.device)
    fn = _lib.lib().dsrg_maxpool3x3_relu_fwd_bf16 if relu_input else _lib.lib().dsrg_maxpool3x3_fwd_bf16
    check(fn(_ptr(x), _ptr(out), _ptr(code), B, H, W, OH, OW, C, stride, _stream()))
    return out, code


def maxpool3x3_bwd(gout, code, in_shape, stride):
    B, C, H, W = in_shape
    OH, OW = gout.shape[2], gout.shape[3]
    gout = gout.contiguous(memory_format=torch.channels_last)
    gin = torch.empty((B, C, H, W), dtype=gout.dtype, device=gout.device, memory_format=torch.channels_last)
    check(_lib.lib().dsrg_maxpool3x3_bwd_bf16(_ptr(gout), _ptr(code), _ptr(gin), B, H, W, OH, OW, C, stride, _stream()))
    return gin


def maxpool3x3_bwd_relu(gout, code, relu_out, stride=2, gb_out=None):
    """3x3 / stride 2 / pad 1 max-pool backward + the ReLU backward and bias gradient of the convolution in front of the pool:
    relu_out (B,C,H,W) bf16 channels_last = the pool's input -> (masked input gradient (B,C,H,W) bf16 channels_last,
    bias gradient (C) f32); the same values as maxpool3x3_bwd followed by relu_bwd_bias.
    relu_out may be the pool input's SHAPE (a torch.Size / tuple) when `code` was made by maxpool3x3_fwd(relu_input=True): the
    codes then carry the mask and the input is not read (same bits)"""
    have_y = torch.is_tensor(relu_out)
    B, C, H, W = relu_out.shape if have_y else relu_out
    OH, OW = gout.shape[2], gout.shape[3]
    cl = torch.channels_last
    if stride != 2 or not (gout.is_cuda and gout.dtype == torch.bfloat16 and C % 8 == 0 and 256 % (C // 8) == 0) or (
            have_y and not (relu_out.is_cuda and relu_out.dtype == torch.bfloat16 and relu_out.is_contiguous(memory_format=cl))):
        raise ValueError("maxpool3x3_bwd_relu: stride 2, bf16 channels_last, channels / 8 a divisor of 256")
    gout = gout.contiguous(memory_format=cl)
    gin = torch.empty((B, C, H, W), dtype=torch.bfloat16, device=gout.device, memory_format=cl)
    gb = _dest(gb_out, (C,), torch.float32, gout.device)
    part = _partial_rows(gout.device, C)                            # shared with relu_bwd_bias (same stream-ordering rule)
    check(_lib.lib().dsrg_maxpool3x3_bwd_relu_bf16(_ptr(gout), _ptr(code), _ptr(relu_out) if have_y else None, _ptr(gin), _ptr(gb), _ptr(part),
                                                   _PARTIAL_BLOCKS, B, H, W, OH, OW, C, _stream()))
    return gin, gb


class DSRGSupervision(torch.autograd.Function):
    """loss-Seed + loss-Constrain as a differentiable function of the fc8 logits."""

    @staticmethod
    def forward(ctx, logits, images, labels, cues, th1, th2, scale_factor, maxiter, prepared=False):
        losses, grad, _ = supervision_step(logits.contiguous(), images, labels, cues, th1, th2, scale_factor, maxiter,
                                           prepared=prepared)
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(losses)
        return losses.sum(), losses

    @staticmethod
    def backward(ctx, g_total, _g_losses):
        (grad,) = ctx.saved_tensors
        return grad * g_total, None, None, None, None, None, None, None, None


def dsrg_supervision_loss(logits, images, labels, cues, th1=0.99, th2=0.85, scale_factor=12.0, maxiter=10,
                          prepared=False):
    """-> (total loss (differentiable wrt logits), tensor [loss-Seed, loss-Constrain])."""
    return DSRGSupervision.apply(logits, images, labels, cues, th1, th2, scale_factor, maxiter, prepared)
