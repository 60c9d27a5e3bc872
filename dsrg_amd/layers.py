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

from . import ops

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


# ---- blobs resident on the device between the layers of one iteration ---------------------------------------------------
# Caffe hands every layer numpy views of its blobs; the protocol forces a host copy of every top.  What it does not force is
# sending the same bytes up again: a blob one of these layers has just written (or uploaded) is still in HBM when the next
# layer receives it as a bottom.  `_dev` therefore keeps, per host buffer (address, shape, dtype), the device tensor together
# with a 64-bit digest (xxh3) of the host bytes it mirrors, and re-uploads only when the host bytes no longer hash to it —
# any write by Caffe or by another layer in between is seen.  Without the xxhash module every call uploads, as before.
# Blobs up to _EXACT_BYTES (every blob of the path but the images: 2.3 MB of probabilities take 0.15 ms) are hashed in full.
# The image blob (20 MB, 1.3 ms per pass — as long as the upload it would save) is fingerprinted by its address, shape, both
# ends and every 16th 64-byte line instead: it is written once per iteration by the data layer, BEFORE the two layers that
# read it, so the question "is this still the buffer CRFLayer saw a moment ago" does not need every byte;
# DSRG_PYLAYERS_EXACT=1 hashes everything.
import os as _os
_EXACT_BYTES = 1 << 62 if _os.environ.get("DSRG_PYLAYERS_EXACT") == "1" else 4 << 20
try:
    import xxhash as _xxhash

    def _digest(a):
        m = memoryview(a).cast("B")
        if m.nbytes <= _EXACT_BYTES:
            return _xxhash.xxh3_64_intdigest(m)
        lines = np.frombuffer(m, dtype=np.uint8, count=(m.nbytes // 64) * 64).reshape(-1, 64)
        h = _xxhash.xxh3_64()
        h.update(m[:4096]); h.update(m[-4096:]); h.update(np.ascontiguousarray(lines[::16]))
        return h.intdigest() ^ m.nbytes
except ImportError:                      # pragma: no cover - the image ships xxhash
    _digest = None

_resident = {}                           # (address, shape, dtype) -> (digest of the host bytes, device tensor)
# DSRG_PYLAYERS_TRUST=1: no digests at all — a resident copy is taken for valid until the next iteration begins
# (SoftmaxLayer.forward, the first layer of the path, train-s.prototxt:746-759).  Right for a Caffe net, whose blobs are written
# only by their producing layer once per iteration; wrong for a driver that rewrites a blob between two of these layers.
_TRUST = _os.environ.get("DSRG_PYLAYERS_TRUST") == "1"
_epoch = [0]


def _key(a):
    return (a.ctypes.data, a.shape, a.dtype.str)


def _dev(a, dtype=torch.float32):
    """the device copy of host array `a` (uploaded, or the resident one when the host bytes are unchanged)"""
    a = np.ascontiguousarray(a)
    if _digest is None or a.dtype != np.float32 or dtype != torch.float32:
        return torch.from_numpy(a).to(device="cuda", dtype=dtype)
    k = _key(a)
    hit = _resident.get(k)
    if _TRUST:
        if hit is not None and hit[0] == ("epoch", _epoch[0]):
            return hit[1]
        d = ("epoch", _epoch[0])
    else:
        d = _digest(a)
        if hit is not None and hit[0] == d:
            return hit[1]
    t = torch.from_numpy(a).to(device="cuda", dtype=dtype)
    _resident[k] = (d, t)
    return t


def _blob_id(a):
    """what identifies a blob's content for the CRF reuse: its digest, or under DSRG_PYLAYERS_TRUST its buffer and the iteration"""
    a = np.ascontiguousarray(a)
    return (_key(a), _epoch[0]) if _TRUST else _digest(a)


def _publish(host, tensor):
    """host[...] = tensor (a top blob, or a bottom clipped in place) and remember that `tensor` mirrors it"""
    host[...] = tensor.detach().cpu().numpy().reshape(host.shape)
    if _digest is not None and host.dtype == np.float32 and tensor.dtype == torch.float32 and host.flags.c_contiguous:
        _resident[_key(host)] = (("epoch", _epoch[0]) if _TRUST else _digest(host), tensor)


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
        _epoch[0] += 1                                       # a new iteration: nothing uploaded before is trusted any more
        _publish(top[0].data, ops.softmax_forward(_dev(bottom[0].data)))

    def backward(self, top, prop_down, bottom):
        grad = ops.softmax_backward(_dev(bottom[0].data), _dev(top[0].diff))
        bottom[0].diff[...] = grad.cpu().numpy()


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
        probs = _dev(bottom[0].data)
        refined, logq = ops.crf_refine(probs, _dev(bottom[1].data), scale_factor=12.0)
        _publish(bottom[0].data, probs)                    # the in-place clip (pylayers.py:65-67)
        self._result_dev = refined
        self.result = refined.cpu().numpy()
        _publish(top[0].data, logq)
        if _digest is not None:
            _last_crf.update(probs=_blob_id(bottom[0].data), images=_blob_id(bottom[1].data), scale=12.0, refined=refined)

    def backward(self, top, prop_down, bottom):
        grad = ops.crf_layer_backward(self._result_dev, _dev(top[0].diff))
        bottom[0].diff[...] = grad.cpu().numpy()


class BalancedSeedLossLayer(_Base):
    """pylayers.py:120-152.  bottom = [probs, seeds]; ignores top.diff like the reference."""

    def setup(self, bottom, top):
        if len(bottom) != 2:
            raise Exception("The layer needs two inputs!")

    def reshape(self, bottom, top):
        top[0].reshape(1)

    def forward(self, bottom, top):
        loss, _ = ops.seed_loss(_dev(bottom[0].data), _dev(bottom[1].data), want_grad=False)
        top[0].data[...] = loss.cpu().numpy()

    def backward(self, top, prop_down, bottom):
        _, grad = ops.seed_loss(_dev(bottom[0].data), _dev(bottom[1].data), want_grad=True)
        bottom[0].diff[...] = grad.cpu().numpy()


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
        loss, _, _ = ops.constrain_loss(_dev(bottom[0].data), _dev(bottom[1].data), want_grad=False)
        top[0].data[...] = loss.cpu().numpy()

    def backward(self, top, prop_down, bottom):
        _, gp, gq = ops.constrain_loss(_dev(bottom[0].data), _dev(bottom[1].data), want_grad=True)
        bottom[0].diff[...] = gp.cpu().numpy()
        bottom[1].diff[...] = gq.cpu().numpy()


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
        refined, _ = ops.crf_refine(p, _dev(im), scale_factor=scale_factor, want_log=False)
        _publish(probs, p)                                    # in-place clip (pylayers.py:312)
        return refined

    def generate_seed(self, labels, probs, cues, im):
        refined = self.refinement(probs, im, 12.0)
        self._seeds_dev = ops.srg_grow(_dev(labels).reshape(labels.shape[0], -1).contiguous(), _dev(cues), refined,
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
