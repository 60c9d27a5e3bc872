"""Caffe Python-layer protocol over the MI355X kernels — the drop-in for the
reference module `pylayers` (pylayers/pylayers/pylayers.py).

Same class names, `setup/reshape/forward/backward(bottom, top)` signatures,
`param_str` keys, bottom orders, side effects and error behaviour as the reference,
so training/experiment/seed_mc/train-s.prototxt:24-39,746-810 binds to these classes
unchanged.  Blobs are anything with `.data` / `.diff` (float32 C-contiguous numpy
arrays, writable in place) and `.reshape(*shape)` — i.e. pycaffe blobs.  Each
forward/backward stages the blob through HBM and runs the HIP kernels of
libdsrg_hip.so; there is no CPU implementation behind these classes.
"""
import os.path as osp
import pickle

import numpy as np
import torch
import yaml

from . import _lib, ops

try:                                    # inside Caffe the layers derive from caffe.Layer (pylayers.py:23)
    import caffe as _caffe
    _Base = _caffe.Layer
except ImportError:                     # stand-alone (tests, PyTorch trainer): same protocol, plain object
    _Base = object

min_prob = 0.0001                        # pylayers.py:20


MAX_LABELS = 96                          # kMaxLabels of libdsrg_hip.so (per-pixel label columns live in registers / LDS)


def _check_labels(n, who):
    """The HIP kernels hold a pixel's label column in registers / LDS: at most 96 labels, which covers the 81-class blobs of
    AnnotationLayerCOCO (pylayers.py:387-507).  More is rejected here, at reshape time, with a message that says so — not
    deep inside a launch."""
    if n > MAX_LABELS:
        raise Exception("%s: %d label planes, but this build of libdsrg_hip.so supports at most %d" % (who, n, MAX_LABELS))


# ---- host blobs <-> HBM --------------------------------------------------------------------------------------------------
# Caffe hands every layer numpy views of its blobs; the protocol forces a host copy of every top and diff.  What it does not
# force, and what this section avoids (round 5; the per-layer cost table is in INTEGRATION.md, tools/pylayers_route_cost.py):
#   * sending the same bytes up again — a blob one of these layers has just written (or uploaded) is still in HBM when the next
#     layer receives it as a bottom: `_dev` keeps, per host buffer (address, shape, dtype), the device tensor together with a
#     64-bit digest (xxh3) of the host bytes it mirrors and re-uploads only when the host bytes no longer hash to it;
#   * hashing the same bytes again and again inside one iteration — a blob that was hashed IN FULL (or written by us) in the
#     current epoch is re-checked by a sampled digest (both ends + every 16th 64-byte line) when the next layer of the same
#     iteration receives it; an epoch begins at SoftmaxLayer.forward (CRFLayer.forward for a driver that starts there: the heads
#     of the path, train-s.prototxt:746-775), so anything written between iterations is hashed in full — except a buffer that
#     held other bytes at each of its last two first sights (the net's scores, a data layer's tops): it is simply uploaded.  DSRG_PYLAYERS_EXACT=1: every
#     check hashes every byte.  The image blob (20 MB: a full pass costs as much as its upload) is sampled as before unless EXACT;
#   * pageable copies — a host buffer seen twice at the same address (Caffe's blobs never move) is page-locked in place
#     (hipHostRegister through torch), so uploads and downloads are single DMA transfers into the blob itself; the registry
#     holds a reference to the array and is bounded (least recently used entries are unregistered and dropped);
#   * one device synchronisation per blob — downloads of a layer call are queued and waited for ONCE, when the call ends;
#   * downloads nobody reads — CRFLayer.result (float64, 4.5 MB per 16 images) is fetched when it is first read.
# Without the xxhash module every call uploads, as before.
import os as _os
import time as _time
from collections import OrderedDict as _OrderedDict

_EXACT = _os.environ.get("DSRG_PYLAYERS_EXACT") == "1"
_EXACT_BYTES = 1 << 62 if _EXACT else 4 << 20
try:
    import xxhash as _xxhash

    def _sampled(m):
        lines = np.frombuffer(m, dtype=np.uint8, count=(m.nbytes // 64) * 64).reshape(-1, 64)
        h = _xxhash.xxh3_64()
        h.update(m[:4096]); h.update(m[-4096:]); h.update(np.ascontiguousarray(lines[::16]))
        return h.intdigest() ^ m.nbytes

    def _digest(a, sampled_ok=False):
        """64-bit digest of a's bytes: all of them, or (sampled_ok, or a blob beyond _EXACT_BYTES) the sampled form"""
        t0 = _time.perf_counter()
        m = memoryview(a).cast("B")
        if m.nbytes < 16384 or (m.nbytes <= _EXACT_BYTES and not (sampled_ok and not _EXACT)):
            d = ("full", _xxhash.xxh3_64_intdigest(m))
        else:
            d = ("sampled", _sampled(m))
        _tick("digest", t0)
        return d
except ImportError:                      # pragma: no cover - the image ships xxhash
    _digest = None

# DSRG_PYLAYERS_TRUST=1: no digests at all — a resident copy is taken for valid until the next iteration begins.  Right for a
# Caffe net, whose blobs are written only by their producing layer once per iteration; wrong for a driver that rewrites a blob
# between two of these layers.
_TRUST = _os.environ.get("DSRG_PYLAYERS_TRUST") == "1"
_epoch = [0]
_MAX_RESIDENT = 64
_resident = _OrderedDict()               # (address, shape, dtype) -> [epoch of the last full check, {form: digest}, device tensor,
                                         #                              misses, epoch in which WE wrote the blob (or -1)]
_pinned = _OrderedDict()                 # (address, nbytes) -> [the array (kept alive), sightings, registered?]
_PIN = _os.environ.get("DSRG_PYLAYERS_PIN", "1") == "1"
_pending = []                            # downloads of the current layer call: (host array, device tensor, remember?)

# ---- cost accounting (tools/pylayers_route_cost.py): seconds per (layer call, category) while `profile` is a dict
profile = None
_call_name = ["?"]


def _tick(cat, t0):
    if profile is not None:
        k = (_call_name[0], cat)
        profile[k] = profile.get(k, 0.0) + (_time.perf_counter() - t0)


def _sync_for(cat):
    """profiling only: wait for the device so that the time spent so far is booked under `cat`"""
    if profile is not None:
        t0 = _time.perf_counter()
        torch.cuda.synchronize()
        _tick(cat, t0)


def _key(a):
    return (a.ctypes.data, a.shape, a.dtype.str)


def _pin(a):
    """page-lock the memory of a C-contiguous host array in place once it has been seen twice at the same address; -> pinned?"""
    if not _PIN or a.nbytes < 65536:
        return False
    k = (a.ctypes.data, a.nbytes)
    e = _pinned.get(k)
    if e is None:
        _pinned[k] = e = [a, 0, False]
        while len(_pinned) > _MAX_RESIDENT:
            _, old = _pinned.popitem(last=False)
            if old[2]:
                _lib.lib().dsrg_host_unregister(old[0].ctypes.data)      # (the entry kept the array alive: still mapped)
    else:
        _pinned.move_to_end(k)
    e[1] += 1
    if not e[2] and e[1] >= 2:
        t0 = _time.perf_counter()
        # inside the library: memory HIP already knows (page-locked by its owner, overlapping an earlier registration) is
        # skipped, and a refused registration leaves no HIP error behind for the next launch check
        e[2] = _lib.lib().dsrg_host_register(a.ctypes.data, a.nbytes) == 1
        if not e[2]:
            e[1] = -(1 << 30)            # do not try again
        _tick("pin", t0)
    return e[2]


_async_uploads = [False]


def _upload(a, dtype):
    t0 = _time.perf_counter()
    t = torch.from_numpy(a)
    pinned = _pin(a)
    _async_uploads[0] = _async_uploads[0] or pinned      # the blob must not change before the DMA has read it: _finish waits
    t = t.to(device="cuda", dtype=dtype, non_blocking=pinned)
    _tick("h2d", t0)
    _sync_for("h2d")
    return t


def _dev(a, dtype=torch.float32, cache=True):
    """the device copy of host array `a` (uploaded, or the resident one when the host bytes are unchanged); cache=False: a
    blob that is consumed once (a diff): uploaded, neither hashed nor kept"""
    c = np.ascontiguousarray(a)
    if _digest is None or c.dtype != np.float32 or dtype != torch.float32 or not cache or c is not a:
        return _upload(c, dtype)         # (a non-contiguous input was copied: its address means nothing)
    k = _key(a)
    hit = _resident.get(k)
    if _TRUST:
        if hit is not None and hit[0] == _epoch[0]:
            return hit[2]
        t = _upload(a, dtype)
        _remember(k, {}, t, 0)
        return t
    misses, full = 0, None
    if hit is not None:
        _resident.move_to_end(k)
        in_epoch = hit[0] == _epoch[0]
        if in_epoch or hit[3] < 2:
            # written by US in this epoch (_publish: nobody else writes a top of ours inside an iteration): the sampled form
            # decides; a blob the host provided is hashed in full at every sight — a sparse host write between two layer calls
            # of one iteration must not be missed (a gradient checker, a unit driver that edits a blob and calls a layer again)
            d = _digest(a, sampled_ok=hit[4] == _epoch[0])
            if hit[1].get(d[0]) == d[1]:
                if d[0] == "full":
                    hit[0], hit[3] = _epoch[0], 0
                return hit[2]
            full = d if d[0] == "full" else None
        # (a buffer that held other bytes at each of its last two first sights — the net's scores, a data layer's tops — is
        # not hashed in full a third time: it is uploaded, which is what the hash would have said)
        misses = hit[3] + (0 if in_epoch else 1)
    t = _upload(a, dtype)
    samp = _digest(a, sampled_ok=True)
    forms = {samp[0]: samp[1]}
    if full is not None:
        forms[full[0]] = full[1]
    elif hit is None or misses < 2 or a.nbytes <= (1 << 20):
        # (a small blob keeps its full digest whatever its history: the images are seen by CRFLayer and again by DSRGLayer in
        # every iteration, and the second sight of a host-provided blob compares every byte)
        full = _digest(a)
        forms[full[0]] = full[1]
    _remember(k, forms, t, misses)
    return t


def _remember(k, forms, t, misses=0, ours=False):
    _resident[k] = [_epoch[0], forms, t, misses, _epoch[0] if ours else -1]
    _resident.move_to_end(k)
    while len(_resident) > _MAX_RESIDENT:
        _resident.popitem(last=False)


def _blob_id(a):
    """what identifies a blob's content for the CRF reuse: its digest, or under DSRG_PYLAYERS_TRUST its buffer and the iteration"""
    a = np.ascontiguousarray(a)
    if _TRUST:
        return (_key(a), _epoch[0])
    hit = _resident.get(_key(a))
    d = _digest(a, sampled_ok=hit is not None and hit[4] == _epoch[0])
    return d


def _publish(host, tensor, remember=True):
    """host[...] = tensor (a top blob, a diff, or a bottom clipped in place), queued: the copies of a layer call are waited for
    once, in _finish(); remember: `tensor` mirrors the blob afterwards (the next layer's _dev finds it)"""
    t0 = _time.perf_counter()
    src = tensor.detach()
    if host.flags.c_contiguous and host.dtype == np.float32 and src.dtype == torch.float32 and _pin(host):
        torch.from_numpy(host).copy_(src.reshape(host.shape), non_blocking=True)      # DMA straight into the blob
        _pending.append((host, None, src if remember else None))
    else:
        stage = src.to("cpu", non_blocking=False) if not src.is_cuda else src.cpu()
        _pending.append((host, stage, src if remember else None))
    _tick("d2h", t0)


def _finish():
    """end of a layer call: wait for the queued downloads, complete the pageable ones, note what now mirrors what"""
    if not _pending and not _async_uploads[0]:
        return
    t0 = _time.perf_counter()
    torch.cuda.current_stream().synchronize()
    _async_uploads[0] = False
    _tick("sync", t0)
    for host, stage, src in _pending:
        if stage is not None:
            t0 = _time.perf_counter()
            host[...] = stage.numpy().reshape(host.shape)
            _tick("d2h", t0)
        if src is not None and _digest is not None and host.dtype == np.float32 and src.dtype == torch.float32 and host.flags.c_contiguous:
            # written by us in this epoch: the sampled form is all a later check of the same iteration compares (a blob that
            # survives into the next epoch is hashed in full there, finds no full digest and is uploaded again)
            forms = {}
            if not _TRUST:
                d = _digest(host, sampled_ok=True)
                forms[d[0]] = d[1]
            _remember(_key(host), forms, src, ours=True)
    del _pending[:]


def _kernels(fn, *a, **kw):
    """fn(*a, **kw) — the HIP kernels of a layer call; under the cost accounting the device is waited for so that their time is
    booked as theirs and not as the next download's"""
    t0 = _time.perf_counter()
    out = fn(*a, **kw)
    _tick("kernels", t0)
    _sync_for("kernels")
    return out


_softmax_ran = [False]
_forwards_seen = set()                   # the forward calls of the current epoch, by name


class _call(object):
    """`with _call("CRFLayer.forward"):` — names the layer call for the cost table and flushes its downloads at the end"""

    def __init__(self, name, new_epoch=False):
        self.name, self.new_epoch = name, new_epoch

    def __enter__(self):
        _call_name[0] = self.name
        # an epoch begins at the head of the path: SoftmaxLayer.forward, or CRFLayer.forward when no SoftmaxLayer ran in front of
        # it (a driver that starts at the CRF) — nothing checked before is taken on a sampled digest any more
        # ... and so does it when a layer's forward runs a SECOND time inside one epoch: a driver that calls a loss layer or
        # DSRGLayer on its own again and again (numeric gradient checks) never passes the head of the path, and the blobs we
        # wrote in "this" iteration may have been edited by it since
        bump = False
        if self.new_epoch == "softmax":
            bump = True
            _softmax_ran[0] = True
        elif self.new_epoch == "crf":
            bump = not _softmax_ran[0]
            _softmax_ran[0] = False
        if self.name.endswith(".forward"):
            if self.name in _forwards_seen:
                bump = True
            if bump:
                _forwards_seen.clear()
            _forwards_seen.add(self.name)
        if bump:
            _epoch[0] += 1
        self.t0 = _time.perf_counter()
        return self

    def __exit__(self, *exc):
        _finish()
        if profile is not None:
            k = (self.name, "total")
            profile[k] = profile.get(k, 0.0) + (_time.perf_counter() - self.t0)
        return False


# the dense CRF of this iteration: CRFLayer.forward and DSRGLayer.refinement run it on the SAME (clipped) probabilities and
# images (train-s.prototxt:760-800; SURVEY §0.2 proved the two results identical) — the second caller takes the first one's
# marginals when both blobs still hash to what the first one saw
_last_crf = {"probs": None, "images": None, "scale": None, "refined": None}
crf_reuse_count = 0                      # how often DSRGLayer.refinement took CRFLayer's marginals (tests, tools)


class SoftmaxLayer(_Base):
    """pylayers.py:23-51"""

    def setup(self, bottom, top):
        if len(bottom) != 1:
            raise Exception("Need two inputs to compute distance.")

    def reshape(self, bottom, top):
        _check_labels(bottom[0].data.shape[1], "SoftmaxLayer")
        top[0].reshape(*bottom[0].data.shape)

    def forward(self, bottom, top):
        with _call("SoftmaxLayer.forward", new_epoch="softmax"):
            _publish(top[0].data, _kernels(ops.softmax_forward, _dev(bottom[0].data)))

    def backward(self, top, prop_down, bottom):
        with _call("SoftmaxLayer.backward"):
            _publish(bottom[0].diff, _kernels(ops.softmax_backward, _dev(bottom[0].data), _dev(top[0].diff, cache=False)), remember=False)


class CRFLayer(_Base):
    """pylayers.py:54-92.  bottom = [probs, images]; clips bottom[0].data in place and
    keeps `self.result` (float64, NCHW) for backward, like the reference."""

    def setup(self, bottom, top):
        if len(bottom) != 2:
            raise Exception("The layer needs two inputs!")

    def reshape(self, bottom, top):
        _check_labels(bottom[0].data.shape[1], "CRFLayer")
        top[0].reshape(*bottom[0].data.shape)

    def forward(self, bottom, top):
        with _call("CRFLayer.forward", new_epoch="crf"):
            probs = _dev(bottom[0].data)
            refined, logq = _kernels(ops.crf_refine, probs, _dev(bottom[1].data), scale_factor=12.0)
            _publish(bottom[0].data, probs)                # the in-place clip (pylayers.py:65-67)
            self._result_dev, self._result_host = refined, None
            _publish(top[0].data, logq)
            _finish()                                      # the blobs hold their final bytes: identify them for DSRGLayer
            if _digest is not None:
                _last_crf.update(probs=_blob_id(bottom[0].data), images=_blob_id(bottom[1].data), scale=12.0, refined=refined)

    @property
    def result(self):
        """the refined marginals, float64 NCHW, as the reference keeps them (pylayers.py:77-86) — downloaded when first read
        (nothing in a Caffe net reads them: backward uses the device copy)"""
        if getattr(self, "_result_host", None) is None and getattr(self, "_result_dev", None) is not None:
            self._result_host = self._result_dev.cpu().numpy()
        return getattr(self, "_result_host", None)

    def backward(self, top, prop_down, bottom):
        with _call("CRFLayer.backward"):
            _publish(bottom[0].diff, _kernels(ops.crf_layer_backward, self._result_dev, _dev(top[0].diff, cache=False)), remember=False)


class BalancedSeedLossLayer(_Base):
    """pylayers.py:120-152.  bottom = [probs, seeds]; ignores top.diff like the reference."""

    def setup(self, bottom, top):
        if len(bottom) != 2:
            raise Exception("The layer needs two inputs!")

    def reshape(self, bottom, top):
        top[0].reshape(1)

    def forward(self, bottom, top):
        with _call("BalancedSeedLossLayer.forward"):
            loss, _ = _kernels(ops.seed_loss, _dev(bottom[0].data), _dev(bottom[1].data), want_grad=False)
            _publish(top[0].data, loss, remember=False)

    def backward(self, top, prop_down, bottom):
        with _call("BalancedSeedLossLayer.backward"):
            _, grad = _kernels(ops.seed_loss, _dev(bottom[0].data), _dev(bottom[1].data), want_grad=True, want_loss=False)
            _publish(bottom[0].diff, grad, remember=False)


class SeedLossLayer(_Base):
    """pylayers.py:94-118 (the unbalanced seeding loss of SEC; no seed_mc prototxt uses it).  bottom = [probs, seeds]."""

    def setup(self, bottom, top):
        if len(bottom) != 2:
            raise Exception("The layer needs two inputs!")

    def reshape(self, bottom, top):
        top[0].reshape(1)

    def forward(self, bottom, top):
        loss, _ = ops.seed_loss_plain(_dev(bottom[0].data), _dev(bottom[1].data), want_grad=False)
        top[0].data[...] = loss.cpu().numpy()

    def backward(self, top, prop_down, bottom):
        _, grad = ops.seed_loss_plain(_dev(bottom[0].data), _dev(bottom[1].data), want_grad=True)
        bottom[0].diff[...] = grad.cpu().numpy()


class ExpandLossLayer(_Base):
    """pylayers.py:183-233 (SEC's expansion loss: global weighted rank pooling with q = 0.996 / 0.999, one sort per
    label plane).  bottom = [probs (B,21,41,41), image-level labels (B,1,1,21)]."""

    def setup(self, bottom, top):
        if len(bottom) != 2:
            raise Exception("The layer needs two inputs!")

    def reshape(self, bottom, top):
        top[0].reshape(1)

    def forward(self, bottom, top):
        loss, _ = ops.expand_loss(_dev(bottom[0].data), _dev(bottom[1].data), want_grad=False)
        top[0].data[...] = loss.cpu().numpy()

    def backward(self, top, prop_down, bottom):
        _, grad = ops.expand_loss(_dev(bottom[0].data), _dev(bottom[1].data), want_grad=True)
        bottom[0].diff[...] = grad.cpu().numpy()


class ConstrainLossLayer(_Base):
    """pylayers.py:154-180.  bottom = [probs, crf_log]."""

    def setup(self, bottom, top):
        if len(bottom) != 2:
            raise Exception("The layer needs two inputs!")

    def reshape(self, bottom, top):
        top[0].reshape(1)

    def forward(self, bottom, top):
        with _call("ConstrainLossLayer.forward"):
            loss, _, _ = _kernels(ops.constrain_loss, _dev(bottom[0].data), _dev(bottom[1].data), want_grad=False)
            _publish(top[0].data, loss, remember=False)

    def backward(self, top, prop_down, bottom):
        with _call("ConstrainLossLayer.backward"):
            _, gp, gq = _kernels(ops.constrain_loss, _dev(bottom[0].data), _dev(bottom[1].data), want_grad=True, want_loss=False)
            _publish(bottom[0].diff, gp, remember=False)
            _publish(bottom[1].diff, gq, remember=False)


class DSRGLayer(_Base):
    """pylayers.py:277-344.  bottom = [labels, probs, cues, images]; param_str is YAML
    with th1 (background threshold), th2 (foreground threshold) and optional iters."""

    def setup(self, bottom, top):
        if len(bottom) != 4:
            raise Exception("The layer needs four inputs!")
        cfg = dict({'iters': -1}, **(yaml.safe_load(self.param_str) or {}))
        self._th1, self._th2, self._max_iters = cfg['th1'], cfg['th2'], cfg['iters']      # th1 / th2 are required (KeyError)
        self._iter_index = 0
        # the reference forks a multiprocessing.Pool here; the batch runs as one kernel launch instead

    def reshape(self, bottom, top):
        _check_labels(bottom[1].data.shape[1], "DSRGLayer")
        top[0].reshape(*bottom[1].data.shape)

    def forward(self, bottom, top):
        with _call("DSRGLayer.forward"):
            img_labels, probs, cues, im = bottom[0].data, bottom[1].data, bottom[2].data, bottom[3].data
            seed_c = self.generate_seed(img_labels, probs, cues, im)
            self._iter_index = self._iter_index + 1
            _publish(top[0].data, seed_c)

    def backward(self, top, prop_down, bottom):
        bottom[1].diff[...] = top[0].diff

    def refinement(self, probs, im, scale_factor=12.0):
        if _digest is not None and _last_crf["refined"] is not None and _last_crf["scale"] == scale_factor and \
                _last_crf["refined"].shape == probs.shape and _last_crf["probs"] == _blob_id(probs) and \
                _last_crf["images"] == _blob_id(im):
            # the blobs CRFLayer.forward refined a moment ago, byte for byte (probs already clipped: the clip is idempotent)
            global crf_reuse_count
            crf_reuse_count += 1
            return _last_crf["refined"]
        p = _dev(probs)
        refined, _ = _kernels(ops.crf_refine, p, _dev(im), scale_factor=scale_factor, want_log=False)
        _publish(probs, p)                                    # in-place clip (pylayers.py:312)
        _finish()                                             # (a public method: the clip is in the blob when it returns)
        return refined

    def generate_seed(self, labels, probs, cues, im):
        refined = self.refinement(probs, im, 12.0)
        self._seeds_dev = _kernels(ops.srg_grow, _dev(labels).reshape(labels.shape[0], -1).contiguous(), _dev(cues), refined,
                                   self._th1, self._th2)
        return self._seeds_dev


def _open_cue_file(name):
    """the localisation-cue dictionary of an AnnotationLayer: a pickle of {'<id>_labels': class ids, '<id>_cues': (3, K)
    int array of (class, row, column) triplets on the 41x41 map}, written by Python 2 (hence latin1).  A relative name is
    looked up where the reference keeps its cue files (pylayers.py:362-363), an absolute path is taken as it is."""
    path = name if osp.isabs(name) else osp.join(osp.dirname(__file__), '../../training', 'localization_cues', name)
    with open(path, 'rb') as f:
        return pickle.load(f, encoding='latin1')


def _mirror_rows(arr, picked):
    """reverse the last axis of arr[i] for the picked i, in place"""
    if picked.any():
        arr[picked] = arr[picked][..., ::-1]


class AnnotationLayer(_Base):
    """Drop-in for pylayers.py:346-387: image ids + images -> image-level labels (B,1,1,21), cue planes (B,21,41,41) and
    the images, with one random horizontal mirror per image applied to cues and image alike.  Host-side marshalling only
    (no arithmetic).  The mirror decisions come from `np.random.choice(2)`, one draw per image in batch order — the
    reference's draws — so a seeded run reproduces the reference's flips (tests/golden/annotation_cases.npz)."""

    num_classes, map_size = 21, 41

    def setup(self, bottom, top):
        if len(bottom) != 2:
            raise Exception("The layer needs two inputs!")
        prm = yaml.safe_load(self.param_str) or {}
        self._cue_name = prm.get('cues', 'localization_cues.pickle')
        self.is_mirror = prm.get('mirror', False)
        self.data_file = _open_cue_file(self._cue_name)

    def reshape(self, bottom, top):
        n = bottom[0].data.shape[0]
        top[0].reshape(n, 1, 1, self.num_classes)
        top[1].reshape(n, self.num_classes, self.map_size, self.map_size)
        top[2].reshape(*bottom[1].data.shape)

    def forward(self, bottom, top):
        ids = [int(v) for v in np.asarray(bottom[0].data).reshape(-1)]
        n = len(ids)
        present = np.zeros((n, self.num_classes), dtype=np.float32)
        present[:, 0] = 1.0                                              # background is present in every image
        planes = np.zeros((n, self.num_classes, self.map_size, self.map_size), dtype=np.float32)
        mirrored = np.zeros(n, dtype=bool)
        for row, image_id in enumerate(ids):
            present[row, self.data_file['%i_labels' % image_id]] = 1.0
            cls, ys, xs = self.data_file['%i_cues' % image_id]
            planes[row, cls, ys, xs] = 1.0
            if self.is_mirror:
                mirrored[row] = np.random.choice(2) == 0                 # the reference's step = choice * 2 - 1: -1 reverses
        images = np.array(bottom[1].data, dtype=np.float32, copy=True)
        _mirror_rows(planes, mirrored)
        _mirror_rows(images, mirrored)
        top[0].data[...] = present.reshape(n, 1, 1, self.num_classes)
        top[1].data[...] = planes
        top[2].data[...] = images

    def backward(self, top, prop_down, bottom):
        pass


class AnnotationLayerCOCO(_Base):
    """Drop-in for pylayers.py:387-507: a self-feeding data layer for the 81-class COCO variant.  `source` lists
    `image_path label_path` pairs under `root`; every forward emits `batch_size` of them: the image resized to `new_size`
    (scipy zoom, order 1, as the reference), RGB, mean-subtracted, CHW; the label PNG as 81 one-hot cue planes (pixels of
    `ignore_label` in none) and as the image-level label vector (1,1,81); one random horizontal mirror for image and cues
    alike; the list is reshuffled whenever it has been walked through.  Host-side marshalling only.  Differences forced by
    this image: PIL instead of the absent OpenCV; `param_str` parsed with ast.literal_eval instead of eval."""

    num_classes = 81

    def setup(self, bottom, top):
        import ast
        prm = ast.literal_eval(self.param_str)
        self.source, self.root_folder, self.batch_size = prm['source'], prm['root'], prm['batch_size']
        self.mean, (self.new_h, self.new_w) = prm['mean'], prm['new_size']
        self.is_mirror = prm.get('mirror', False)
        self.ignore_label = prm.get('ignore_label', 255)
        with open(self.source) as f:
            self.indexlist = [ln.split() for ln in f if ln.strip()]
        self._cur = 0
        top[0].reshape(self.batch_size, 1, 1, self.num_classes)
        top[1].reshape(self.batch_size, self.num_classes, self.new_h // 8 + 1, self.new_w // 8 + 1)
        top[2].reshape(self.batch_size, 3, self.new_h, self.new_w)

    def reshape(self, bottom, top):
        pass

    def backward(self, top, propagate_down, bottom):
        pass

    def forward(self, bottom, top):
        for slot in range(self.batch_size):
            top[2].data[slot, ...], top[1].data[slot, ...], top[0].data[slot, ...] = self.load_next_image()

    def load_next_image(self):
        """-> (image (3,new_h,new_w), cue planes (81,h,w), image-level labels (1,1,81)) of the next list entry"""
        import random
        from .data import _imread_bgr, _imread_gray
        if self._cur >= len(self.indexlist):                               # an epoch is through: new order
            random.shuffle(self.indexlist)
            self._cur = 0
        image_name, label_name = self.indexlist[self._cur]
        self._cur += 1
        return self.preprocess(_imread_bgr(self.root_folder + image_name), _imread_gray(self.root_folder + label_name))

    def preprocess(self, image, label):
        from scipy.ndimage import zoom
        bgr = np.asarray(image, dtype=np.float32)
        bgr = zoom(bgr, (self.new_h / float(bgr.shape[0]), self.new_w / float(bgr.shape[1]), 1.0), order=1)
        chw = np.transpose(bgr[:, :, ::-1] - self.mean, (2, 0, 1))        # BGR -> RGB, mean, channels first
        kept = label != self.ignore_label
        planes = np.zeros((self.num_classes,) + label.shape, dtype=np.uint8)
        rows, cols = np.nonzero(kept)
        planes[label[rows, cols], rows, cols] = 1                          # one scatter instead of the per-pixel loop
        if self.is_mirror and np.random.choice(2) == 0:
            chw, planes = chw[:, :, ::-1], planes[:, :, ::-1]
        present = np.zeros((1, 1, self.num_classes))
        present[0, 0, np.unique(label[kept])] = 1
        return chw, planes, present
