"""ctypes front-end of the CPU oracle (oracle/dsrg_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Never imported by the product packages
(dsrg_amd, pylayers, krahenbuhl2013).  Parity status of each part is stated in
the header of dsrg_oracle.c (SRG/CC pinned against the reference Python; CRF
and the Theano layers unpinned).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_float_p = ctypes.POINTER(ctypes.c_float)
c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_short_p = ctypes.POINTER(ctypes.c_short)
c_ubyte_p = ctypes.POINTER(ctypes.c_ubyte)


def build():
    """Compile liboracle.so with gcc (idempotent).  DSRG_ORACLE_LIB names another build of the same source to load instead
    (tests/test_oracle_sanitized.py: the -fsanitize=address,undefined build)."""
    if os.environ.get("DSRG_ORACLE_LIB"):
        return os.environ["DSRG_ORACLE_LIB"]
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "dsrg_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.orc_crf_create.restype = ctypes.c_void_p
        L.orc_crf_create.argtypes = [ctypes.c_int] * 3
        L.orc_crf_destroy.argtypes = [ctypes.c_void_p]
        L.orc_crf_set_unary_energy.argtypes = [ctypes.c_void_p, c_float_p]
        L.orc_crf_add_pairwise_energy.argtypes = [ctypes.c_void_p] + [ctypes.c_float] * 9 + [c_ubyte_p]
        L.orc_crf_inference.argtypes = [ctypes.c_void_p, ctypes.c_int, c_float_p]
        L.orc_crf_map.argtypes = [ctypes.c_void_p, ctypes.c_int, c_int_p]
        L.orc_crf_lattice_size.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_crf_lattice_size.restype = ctypes.c_int
        L.orc_crf_lattice_norm.argtypes = [ctypes.c_void_p, ctypes.c_int, c_float_p]
        L.orc_crf_lattice_dump.argtypes = [ctypes.c_void_p, ctypes.c_int, c_short_p, c_int_p, c_float_p]
        L.orc_crf_lattice_neighbours.argtypes = [ctypes.c_void_p, ctypes.c_int, c_int_p, c_int_p]
        L.orc_crf_lattice_filter.argtypes = [ctypes.c_void_p, ctypes.c_int, c_float_p, c_float_p, ctypes.c_int]
        L.orc_crf_kernel_filter.argtypes = [ctypes.c_void_p, ctypes.c_int, c_float_p, c_float_p, ctypes.c_int]
        L.orc_crf_refine_batch.argtypes = [ctypes.c_int] * 4 + [c_float_p, c_float_p, ctypes.c_int, ctypes.c_int,
                                                              ctypes.c_double, ctypes.c_int, c_double_p, c_float_p]
        L.orc_crf_layer_backward.argtypes = [ctypes.c_size_t, c_double_p, c_float_p, c_float_p]
        L.orc_cc_label8.argtypes = [c_int_p, ctypes.c_int, ctypes.c_int, c_int_p]
        L.orc_srg_grow.argtypes = [ctypes.c_int] * 3 + [c_float_p, c_float_p, c_double_p, ctypes.c_double, ctypes.c_double]
        L.orc_srg_grow_batch.argtypes = [ctypes.c_int] * 4 + [c_float_p, c_float_p, c_double_p,
                                                            ctypes.c_double, ctypes.c_double, c_float_p]
        L.orc_softmax_forward.argtypes = [ctypes.c_int] * 3 + [c_float_p, c_float_p]
        L.orc_softmax_backward.argtypes = [ctypes.c_int] * 3 + [c_float_p, c_float_p, c_float_p]
        L.orc_seed_loss.argtypes = [ctypes.c_int] * 3 + [c_float_p, c_float_p, c_float_p]
        L.orc_seed_loss.restype = ctypes.c_double
        L.orc_seed_loss_plain.argtypes = [ctypes.c_int] * 3 + [c_float_p, c_float_p, c_float_p]
        L.orc_seed_loss_plain.restype = ctypes.c_double
        L.orc_expand_loss.argtypes = [ctypes.c_int] * 3 + [c_float_p, c_float_p, ctypes.c_double, ctypes.c_double, c_float_p]
        L.orc_expand_loss.restype = ctypes.c_double
        L.orc_confusion_matrix.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_ubyte), ctypes.POINTER(ctypes.c_ubyte),
                                           ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        L.orc_confusion_matrix.restype = None
        L.orc_constrain_loss.argtypes = [ctypes.c_int] * 3 + [c_float_p, c_float_p, c_float_p, c_float_p]
        L.orc_constrain_loss.restype = ctypes.c_double
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class DenseCRF(object):
    """Restatement of the Cython class krahenbuhl2013.wrapper.DenseCRF (wrapper.pyx:20-60)."""

    def __init__(self, W, H, nlabels):
        self.W, self.H, self.M = int(W), int(H), int(nlabels)
        self._h = lib().orc_crf_create(self.W, self.H, self.M)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_crf_destroy(self._h)
            self._h = None

    def set_unary_energy(self, unary_costs):
        u = _f32(unary_costs).ravel()
        assert u.size == self.W * self.H * self.M
        lib().orc_crf_set_unary_energy(self._h, _p(u, c_float_p))

    def add_pairwise_energy(self, w1, ta1, ta2, tb1, tb2, tb3, w2, tg1, tg2, im):
        im = np.ascontiguousarray(im, dtype=np.uint8).ravel()
        assert im.size == self.W * self.H * 3
        lib().orc_crf_add_pairwise_energy(self._h, w1, ta1, ta2, tb1, tb2, tb3, w2, tg1, tg2, _p(im, c_ubyte_p))

    def inference(self, n_iters=10):
        out = np.empty(self.W * self.H * self.M, dtype=np.float32)
        lib().orc_crf_inference(self._h, int(n_iters), _p(out, c_float_p))
        return out

    def map(self, n_iters=10):
        out = np.empty(self.W * self.H, dtype=np.int32)
        lib().orc_crf_map(self._h, int(n_iters), _p(out, c_int_p))
        return out

    # --- introspection for tests ---
    def lattice_size(self, k):
        return lib().orc_crf_lattice_size(self._h, k)

    def lattice_norm(self, k):
        out = np.empty(self.W * self.H, dtype=np.float32)
        lib().orc_crf_lattice_norm(self._h, k, _p(out, c_float_p))
        return out

    def lattice_dump(self, k):
        d = 2 if k == 0 else 5
        M, N = self.lattice_size(k), self.W * self.H
        keys = np.empty((M, d), dtype=np.int16)
        off = np.empty((N, d + 1), dtype=np.int32)
        bary = np.empty((N, d + 1), dtype=np.float32)
        lib().orc_crf_lattice_dump(self._h, k, _p(keys, c_short_p), _p(off, c_int_p), _p(bary, c_float_p))
        return keys, off, bary

    def lattice_neighbours(self, k):
        """blur neighbour ids (d+1, M) x 2, -1 = none (permutohedral.cpp:303-318)"""
        d = 2 if k == 0 else 5
        M = self.lattice_size(k)
        n1 = np.empty((d + 1, M), dtype=np.int32)
        n2 = np.empty((d + 1, M), dtype=np.int32)
        lib().orc_crf_lattice_neighbours(self._h, k, _p(n1, c_int_p), _p(n2, c_int_p))
        return n1, n2

    def lattice_filter(self, k, x):
        """x: (N, vs) float32 (label-fastest) -> filtered, same shape."""
        x = _f32(x)
        out = np.empty_like(x)
        lib().orc_crf_lattice_filter(self._h, k, _p(x, c_float_p), _p(out, c_float_p), x.shape[1])
        return out

    def kernel_filter(self, k, x):
        """DenseKernel::filter (pairwise.cpp:63-80): norm . K (norm . x); x (N, vs) float32 label-fastest."""
        x = _f32(x)
        out = np.empty_like(x)
        lib().orc_crf_kernel_filter(self._h, k, _p(x, c_float_p), _p(out, c_float_p), x.shape[1])
        return out


class lattice_path(object):
    """context manager: lattices built inside use the reference's SCALAR Permutohedral::init (permutohedral.cpp:323-474)
    instead of the SSE one (:140-321) — a cross-check of the two restatements, never used by a parity test"""

    def __init__(self, scalar=True):
        self.scalar = scalar

    def __enter__(self):
        lib().orc_set_lattice_path(int(bool(self.scalar)))
        return self

    def __exit__(self, *exc):
        lib().orc_set_lattice_path(0)
        return False


def CRF(image, unary, maxiter=10, scale_factor=1.0, color_factor=13):
    """Restatement of krahenbuhl2013.CRF (CRF/krahenbuhl2013/CRF.py:4-37)."""
    assert image.shape[:2] == unary.shape[:2]
    H, W = image.shape[:2]
    nlabels = unary.shape[2]
    crf = DenseCRF(W, H, nlabels)
    crf.set_unary_energy(-unary.ravel().astype('float32'))
    crf.add_pairwise_energy(10, 80 / scale_factor, 80 / scale_factor, color_factor, color_factor, color_factor,
                            3, 3 / scale_factor, 3 / scale_factor, image.ravel().astype('ubyte'))
    return crf.inference(maxiter).reshape((H, W, nlabels))


def crf_refine_batch(probs, images, scale_factor=12.0, maxiter=10):
    """CRFLayer.forward / DSRGLayer.refinement (pylayers.py:63-88,310-331).
    probs (B,C,H,W) f32 is clipped IN PLACE; returns (refined f64, logq f32)."""
    assert probs.dtype == np.float32 and probs.flags.c_contiguous
    images = _f32(images)
    B, C, H, W = probs.shape
    refined = np.empty((B, C, H, W), dtype=np.float64)
    logq = np.empty((B, C, H, W), dtype=np.float32)
    lib().orc_crf_refine_batch(B, C, H, W, _p(probs, c_float_p), _p(images, c_float_p),
                               images.shape[2], images.shape[3], float(scale_factor), int(maxiter),
                               _p(refined, c_double_p), _p(logq, c_float_p))
    return refined, logq


def crf_layer_backward(refined, top_diff):
    refined = np.ascontiguousarray(refined, dtype=np.float64)
    top_diff = _f32(top_diff)
    out = np.empty(refined.shape, dtype=np.float32)
    lib().orc_crf_layer_backward(refined.size, _p(refined, c_double_p), _p(top_diff, c_float_p), _p(out, c_float_p))
    return out


def cc_label8(mat):
    mat = np.ascontiguousarray(mat, dtype=np.int32)
    out = np.empty_like(mat)
    lib().orc_cc_label8(_p(mat, c_int_p), mat.shape[0], mat.shape[1], _p(out, c_int_p))
    return out


def srg_grow(labels, seed, refined, th1, th2):
    """generate_seed_step (pylayers.py:237-275) for one image; returns the grown seeds."""
    seed = _f32(seed).copy()
    labels = _f32(labels).ravel()
    refined = np.ascontiguousarray(refined, dtype=np.float64)
    C, H, W = seed.shape
    lib().orc_srg_grow(C, H, W, _p(labels, c_float_p), _p(seed, c_float_p), _p(refined, c_double_p),
                       float(th1), float(th2))
    return seed


def srg_grow_batch(labels, cues, refined, th1=0.99, th2=0.85):
    cues = _f32(cues)
    labels = _f32(labels)
    refined = np.ascontiguousarray(refined, dtype=np.float64)
    B, C, H, W = cues.shape
    out = np.empty_like(cues)
    lib().orc_srg_grow_batch(B, C, H, W, _p(labels, c_float_p), _p(cues, c_float_p), _p(refined, c_double_p),
                             float(th1), float(th2), _p(out, c_float_p))
    return out


def softmax_forward(x):
    x = _f32(x)
    B, C, H, W = x.shape
    p = np.empty_like(x)
    lib().orc_softmax_forward(B, C, H * W, _p(x, c_float_p), _p(p, c_float_p))
    return p


def softmax_backward(x, g):
    x, g = _f32(x), _f32(g)
    B, C, H, W = x.shape
    dx = np.empty_like(x)
    lib().orc_softmax_backward(B, C, H * W, _p(x, c_float_p), _p(g, c_float_p), _p(dx, c_float_p))
    return dx


def seed_loss(p, S, want_grad=True):
    p, S = _f32(p), _f32(S)
    B, C, H, W = p.shape
    g = np.empty_like(p) if want_grad else None
    loss = lib().orc_seed_loss(B, C, H * W, _p(p, c_float_p), _p(S, c_float_p),
                               _p(g, c_float_p) if want_grad else None)
    return loss, g


def seed_loss_plain(p, S, want_grad=True):
    """SeedLossLayer (pylayers.py:94-118)"""
    p, S = _f32(p), _f32(S)
    B, C, H, W = p.shape
    g = np.empty_like(p) if want_grad else None
    loss = lib().orc_seed_loss_plain(B, C, H * W, _p(p, c_float_p), _p(S, c_float_p),
                                     _p(g, c_float_p) if want_grad else None)
    return loss, g


def expand_loss(p, stat, want_grad=True, q_fg=0.996, q_bg=0.999):
    """ExpandLossLayer (pylayers.py:183-233): p (B,C,H,W), stat (B,1,1,C)"""
    p, stat = _f32(p), _f32(stat)
    B, C, H, W = p.shape
    assert stat.size == B * C
    g = np.empty_like(p) if want_grad else None
    loss = lib().orc_expand_loss(B, C, H * W, _p(p, c_float_p), _p(stat, c_float_p), q_fg, q_bg,
                                 _p(g, c_float_p) if want_grad else None)
    return loss, g


def confusion_matrix(gt, pred, nclass, rule_lt=False):
    """evaluate.py:25-30 (rule gt != 255) / :61-68 (rule gt < nclass)"""
    gt = np.ascontiguousarray(gt, dtype=np.uint8).ravel()
    pred = np.ascontiguousarray(pred, dtype=np.uint8).ravel()
    M = np.zeros((nclass, nclass), dtype=np.float64)
    lib().orc_confusion_matrix(gt.size, _p(gt, ctypes.POINTER(ctypes.c_ubyte)), _p(pred, ctypes.POINTER(ctypes.c_ubyte)),
                               nclass, int(rule_lt), _p(M, ctypes.POINTER(ctypes.c_double)))
    return M


def constrain_loss(p, lq, want_grad=True):
    p, lq = _f32(p), _f32(lq)
    B, C, H, W = p.shape
    gp = np.empty_like(p) if want_grad else None
    gq = np.empty_like(p) if want_grad else None
    loss = lib().orc_constrain_loss(B, C, H * W, _p(p, c_float_p), _p(lq, c_float_p),
                                    _p(gp, c_float_p) if want_grad else None,
                                    _p(gq, c_float_p) if want_grad else None)
    return loss, gp, gq
