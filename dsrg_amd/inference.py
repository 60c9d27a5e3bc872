"""Test-time path of the reference on MI355X ("next" row 1): multi-scale inference,
full-resolution dense CRF on log-probabilities, pseudo-label generation, and the evaluation metrics.

  preprocess            <-> training/tools/test-ms.py:68-81
  predict_mask_ms       <-> training/tools/test-ms.py:84-111        (run.sh:6,10: pseudo labels / final test)
  predict_train_gt      <-> training/tools/generate_train_gt.py:78-106
  ConfusionMatrix       <-> training/tools/evaluate.py:17-68

The network runs in PyTorch-ROCm; resampling uses align-corners bilinear interpolation, which is the
sampling scipy.ndimage.zoom(order=1) performs ((in-1)/(out-1) mapping); the CRF is
krahenbuhl2013.CRF (device-resident form crf.CRF_device) -> libdsrg_hip.so (global-memory lattice path for full-resolution maps).
"""
import contextlib

import numpy as np
import torch
import torch.nn.functional as F

from .crf import CRF_device

MEAN_PIXEL = (104.0, 117.0, 123.0)


def _zoom(x, h, w):
    """x: (B,C,h0,w0) -> (B,C,h,w), order-1 zoom with scipy's (in-1)/(out-1) coordinate mapping"""
    if x.shape[2] == h and x.shape[3] == w:
        return x
    # in float64: scipy evaluates the sampling coordinate o * (in - 1) / (out - 1) and the two-tap blend in double and rounds
    # once; with float32 weights the result is off by up to 4e-5 of the value range (measured against scipy.ndimage.zoom,
    # tests/test_inference.py), in float64 by the final rounding only
    return F.interpolate(x.double(), size=(h, w), mode="bilinear", align_corners=True).to(x.dtype)


@contextlib.contextmanager
def _eval_mode(net):
    """the reference runs these passes with caffe.TEST, where Dropout is the identity (test-ms.py:56-58): switch a net
    that comes straight from a trainer to eval mode for the call and restore its mode afterwards"""
    was_training = net.training
    net.eval()
    try:
        yield net
    finally:
        net.train(was_training)


def preprocess(image, size, device="cuda"):
    """image: (H,W,3) RGB uint8/float -> (1,3,size,size) float32 BGR, mean-subtracted (test-ms.py:68-81)"""
    # (uploaded in the image's own dtype and converted on the device: a float32 conversion on the host is a multi-threaded torch op,
    # and the idle spin of its ~100 worker threads after every call eats a container's CPU quota — 7 ms per image became 95)
    x = torch.from_numpy(np.ascontiguousarray(np.asarray(image))).to(device).to(torch.float32).permute(2, 0, 1)[None]
    x = _zoom(x, size, size)
    x = x[:, [2, 1, 0]]
    return x - torch.tensor(MEAN_PIXEL, dtype=torch.float32, device=device).view(1, 3, 1, 1)


class GraphedForward(object):
    """`net(x)` of an eval-mode network for the FIXED input shapes of the test-time loop (test-ms.py resizes every image to 241 / 321 /
    401 before the forward, whatever its own size) as captured HIP graphs: one backbone.GraphedForward per input shape (and autocast
    state), made at the first call with that shape and replayed afterwards.  A batch-1 forward is ~40 launches of 10-30 us of GPU
    work each; issued one by one from Python it takes 1.4 ms whatever the map size (host-bound), replayed it takes what the GPU needs.
    The graph re-reads (and re-packs) the parameters at every replay, so in-place weight updates are seen; the returned tensor is the
    graph's static output: consume it before the next call with the same shape."""

    def __init__(self, net):
        self.net = net
        self._g = {}

    def __call__(self, x):
        amp = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else None
        key = (tuple(x.shape), x.dtype, amp)
        g = self._g.get(key)
        if g is None:
            from .backbone import GraphedForward as _OneShape
            if self.net.training:
                raise RuntimeError("GraphedForward needs an eval-mode network (no dropout stream inside a graph)")
            g = self._g[key] = _OneShape(self.net, x, amp_dtype=amp)
        return g(x)


@torch.no_grad()
def multiscale_scores(net, image, sizes=(241, 321, 401), device="cuda", forward=None):
    """sum over scales of the fc8 scores zoomed to the image resolution (test-ms.py:89-97) -> (C,H,W).  forward: a callable used in
    place of `net` for the forward passes (a GraphedForward of the same net, already in eval mode)"""
    d1, d2 = image.shape[0], image.shape[1]
    total = None
    with _eval_mode(net):
        for size in sizes:
            scores = (forward or net)(preprocess(image, size, device)).float()
            scores = _zoom(scores, d1, d2)
            total = scores if total is None else total + scores
    return total[0]


def _probs_from_scores(scores, eps=0.00001):
    probs = torch.softmax(scores, dim=0)
    return torch.clamp(probs, min=eps)                       # probs[probs < eps] = eps (test-ms.py:102-103)


@torch.no_grad()
def predict_mask_ms(net, image, smooth=True, sizes=(241, 321, 401), device="cuda", forward=None):
    """test-ms.py:84-111 -> (H,W) int64 label mask.  forward: see multiscale_scores"""
    probs = _probs_from_scores(multiscale_scores(net, image, sizes, device, forward))
    if smooth:
        # scores, log-probabilities, CRF and arg-max all stay on the GPU; only the (H,W) mask crosses PCIe
        unary = torch.log(probs).permute(1, 2, 0).contiguous()
        img = torch.as_tensor(np.asarray(image).astype('ubyte'), device=unary.device)
        return CRF_device(img, unary, scale_factor=1.0, want="map").cpu().numpy().astype(np.int64)
    return probs.argmax(0).cpu().numpy()


@torch.no_grad()
def predict_masks_ms_many(net, images, sizes=(241, 321, 401), device="cuda", forward=None, in_flight=3, batch=1):
    """predict_mask_ms over many images (the loop of test-ms.py over a split's 1 449 / 10 582 images), as a generator of (H,W) int64
    masks in order: the forwards of image i + 1 (on the caller's stream; `forward`: a GraphedForward) run while the CRFs of the images
    before it are in flight on `in_flight` worker streams (crf.CRF_device_many; batch > 1: consecutive same-sized images share one
    batched CRF call).  Same masks as predict_mask_ms image by image."""
    from .crf import CRF_device_many

    def pairs():
        for image in images:
            probs = _probs_from_scores(multiscale_scores(net, image, sizes, device, forward))
            unary = torch.log(probs).permute(1, 2, 0).contiguous()
            yield torch.as_tensor(np.asarray(image).astype('ubyte'), device=unary.device), unary

    # the CRF workers come back from the library three times per image and need the interpreter lock for a few lines each time; the
    # caller's thread, busy issuing torch ops, would keep it for Python's default 5 ms switch interval — longer than a whole CRF
    import sys
    interval = sys.getswitchinterval()
    sys.setswitchinterval(min(interval, 1e-4))
    # ... and the forwards go to a stream of their own: on the caller's (usually the null) stream they share a hardware queue with one of
    # the CRF workers' streams or not, depending on how many streams the process made before — 178 or 240 images/s from run to run
    import os
    fstream = None if os.environ.get("DSRG_TEST_MS_FSTREAM", "1") == "0" else torch.cuda.Stream(device=device)      # 0: tools, A/B
    try:
        if fstream is not None:
            fstream.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(fstream):
            for lab in CRF_device_many(pairs(), scale_factor=1.0, want="map", in_flight=in_flight, batch=batch):
                yield lab.cpu().numpy().astype(np.int64)
    finally:
        sys.setswitchinterval(interval)


@torch.no_grad()
def predict_train_gt(net, image, labels, smooth=True, device="cuda"):
    """generate_train_gt.py:78-106: single-scale (321) softmax, zoomed to the image, CRF on log-probs,
    argmax restricted to background + the image-level labels -> (H,W) int64 pseudo-label mask"""
    d1, d2 = image.shape[0], image.shape[1]
    with _eval_mode(net):
        scores = net(preprocess(image, 321, device)).float()
    probs = _zoom(torch.softmax(scores, dim=1), d1, d2)[0]
    probs = torch.clamp(probs, min=0.00001)
    if smooth:
        unary = torch.log(probs).permute(1, 2, 0).contiguous()
        img = torch.as_tensor(np.asarray(image).astype('ubyte'), device=unary.device)
        p = CRF_device(img, unary, scale_factor=1.0)
    else:
        p = probs.permute(1, 2, 0)
    sel = torch.as_tensor([0] + [int(l) for l in labels], device=p.device)
    return sel[p[:, :, sel].argmax(2)].cpu().numpy()


class ConfusionMatrix(object):
    """evaluate.py:17-68 (rows = ground truth, columns = prediction, 255 ignored)."""

    def __init__(self, nclass, classes=None):
        self.nclass = nclass
        self.classes = classes
        self.M = np.zeros((nclass, nclass))

    def add(self, gt, pred):
        gt, pred = np.asarray(gt).ravel(), np.asarray(pred).ravel()
        assert np.max(pred) <= self.nclass
        assert len(gt) == len(pred)
        keep = gt != 255
        self.M += np.bincount(gt[keep].astype(np.int64) * self.nclass + pred[keep].astype(np.int64),
                              minlength=self.nclass ** 2).reshape(self.nclass, self.nclass)

    def add_device(self, gt, pred):
        """`add` for uint8 CUDA tensors: the histogram runs on the GPU (dsrg_confusion_matrix), M is updated on the host"""
        from . import ops
        h = ops.confusion_matrix(gt.reshape(-1), pred.reshape(-1), self.nclass).cpu().numpy()
        assert h[-1] == 0, "labels or predictions outside [0, nclass)"
        self.M += h[:-1].reshape(self.nclass, self.nclass).astype(np.float64)

    def generateM(self, item):
        """evaluate.py:61-68: the matrix of one (gt, pred) pair, keeping ground truth < nclass"""
        gt, pred = np.asarray(item[0]).ravel(), np.asarray(item[1]).ravel()
        assert len(gt) == len(pred)
        keep = gt < self.nclass
        return np.bincount(gt[keep].astype(np.int64) * self.nclass + pred[keep].astype(np.int64),
                           minlength=self.nclass ** 2).reshape(self.nclass, self.nclass).astype(np.float64)

    def addM(self, matrix):
        assert matrix.shape == self.M.shape
        self.M += matrix

    def recall(self):
        with np.errstate(divide="ignore", invalid="ignore"):
            return float(np.sum(np.diag(self.M) / np.sum(self.M, axis=0)) / self.nclass)

    def accuracy(self):
        with np.errstate(divide="ignore", invalid="ignore"):
            return float(np.sum(np.diag(self.M) / np.sum(self.M, axis=1)) / self.nclass)

    def jaccard(self):
        d = np.diag(self.M)
        per = [d[i] / (np.sum(self.M[i, :]) + np.sum(self.M[:, i]) - d[i]) for i in range(self.nclass) if d[i] != 0]
        return np.sum(per) / len(per), per, self.M
