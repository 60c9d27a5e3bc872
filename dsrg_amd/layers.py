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


def _dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device="cuda", dtype=dtype)


class SoftmaxLayer(_Base):
    """pylayers.py:23-51"""

    def setup(self, bottom, top):
        if len(bottom) != 1:
            raise Exception("Need two inputs to compute distance.")

    def reshape(self, bottom, top):
        _check_labels(bottom[0].data.shape[1], "SoftmaxLayer")
        top[0].reshape(*bottom[0].data.shape)

    def forward(self, bottom, top):
        top[0].data[...] = ops.softmax_forward(_dev(bottom[0].data)).cpu().numpy()

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
        bottom[0].data[...] = probs.cpu().numpy()          # the in-place clip (pylayers.py:65-67)
        self._result_dev = refined
        self.result = refined.cpu().numpy()
        top[0].data[...] = logq.cpu().numpy()

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
        layer_params = yaml.safe_load(self.param_str)
        self._th1 = layer_params['th1']
        self._th2 = layer_params['th2']
        if 'iters' not in layer_params:
            layer_params['iters'] = -1
        self._max_iters = layer_params['iters']
        self._iter_index = 0
        # the reference forks a multiprocessing.Pool here; the batch runs as one kernel launch instead

    def reshape(self, bottom, top):
        _check_labels(bottom[1].data.shape[1], "DSRGLayer")
        top[0].reshape(*bottom[1].data.shape)

    def forward(self, bottom, top):
        img_labels, probs, cues, im = bottom[0].data, bottom[1].data, bottom[2].data, bottom[3].data
        seed_c = self.generate_seed(img_labels, probs, cues, im)
        self._iter_index = self._iter_index + 1
        top[0].data[...] = seed_c

    def backward(self, top, prop_down, bottom):
        bottom[1].diff[...] = top[0].diff

    def refinement(self, probs, im, scale_factor=12.0):
        p = _dev(probs)
        refined, _ = ops.crf_refine(p, _dev(im), scale_factor=scale_factor, want_log=False)
        probs[...] = p.cpu().numpy()                          # in-place clip (pylayers.py:312)
        return refined

    def generate_seed(self, labels, probs, cues, im):
        refined = self.refinement(probs, im, 12.0)
        seeds = ops.srg_grow(_dev(labels).reshape(labels.shape[0], -1).contiguous(), _dev(cues), refined,
                             self._th1, self._th2)
        return seeds.cpu().numpy()


class AnnotationLayer(_Base):
    """pylayers.py:346-387: image ids -> labels (B,1,1,21), cues (B,21,41,41), images, with one
    random horizontal flip per image applied to cues and image alike.  Pure data marshalling
    (no arithmetic), done on the host exactly as in the reference.  The cue pickle
    ('%i_labels' -> class ids, '%i_cues' -> (c,h,w) index triplets) is looked up relative to
    this file like the reference does, or at the absolute path given in `cues`."""

    def setup(self, bottom, top):
        if len(bottom) != 2:
            raise Exception("The layer needs two inputs!")
        layer_params = yaml.safe_load(self.param_str)
        if 'cues' not in layer_params:
            layer_params['cues'] = 'localization_cues.pickle'
        self._cue_name = layer_params['cues']
        if 'mirror' not in layer_params:
            layer_params['mirror'] = False
        self.is_mirror = layer_params['mirror']
        path = self._cue_name
        if not osp.isabs(path):
            path = osp.join(osp.dirname(__file__), '../../training', 'localization_cues', self._cue_name)
        with open(path, 'rb') as f:
            self.data_file = pickle.load(f, encoding='latin1')     # py2 cPickle files

    def reshape(self, bottom, top):
        top[0].reshape(bottom[0].data.shape[0], 1, 1, 21)
        top[1].reshape(bottom[0].data.shape[0], 21, 41, 41)
        top[2].reshape(*bottom[1].data.shape)

    def forward(self, bottom, top):
        top[0].data[...] = 0.0
        top[1].data[...] = 0.0
        top[2].data[...] = bottom[1].data
        for i, image_id in enumerate(bottom[0].data[...].ravel()):
            labels_i = self.data_file['%i_labels' % image_id]
            top[0].data[i, 0, 0, 0] = 1.0
            top[0].data[i, 0, 0, labels_i] = 1.0
            cues_i = self.data_file['%i_cues' % image_id]
            top[1].data[i, cues_i[0], cues_i[1], cues_i[2]] = 1.0
            if self.is_mirror:
                flip = np.random.choice(2) * 2 - 1
                top[1].data[i, ...] = top[1].data[i, :, :, ::flip]
                top[2].data[i, ...] = top[2].data[i, :, :, ::flip]

    def backward(self, top, prop_down, bottom):
        pass


class AnnotationLayerCOCO(_Base):
    """pylayers.py:387-507: a self-feeding data layer for the 81-class COCO variant.  `source` lists
    `image_path label_path` pairs under `root`; every forward loads `batch_size` pairs, resizes the image to
    `new_size` (bilinear, scipy zoom order 1 as the reference), RGB, mean-subtracted, CHW; the label PNG becomes
    81 one-hot cue planes (ignore_label skipped) and the image-level label vector (1,1,81); one random horizontal
    flip for image and cues alike; the list is reshuffled at every epoch end.  Host-side marshalling only.
    Differences forced by the image: PIL instead of the absent OpenCV; `param_str` parsed with ast.literal_eval
    instead of eval; the per-pixel Python loop over the label is one vectorised scatter."""

    num_classes = 81

    def setup(self, bottom, top):
        import ast
        layer_params = ast.literal_eval(self.param_str)
        self.source = layer_params['source']
        self.root_folder = layer_params['root']
        self.batch_size = layer_params['batch_size']
        self.is_mirror = layer_params.get('mirror', False)
        self.mean = layer_params['mean']
        self.new_h, self.new_w = layer_params['new_size']
        self.ignore_label = layer_params.get('ignore_label', 255)
        with open(self.source) as f:
            self.indexlist = [line.strip().split() for line in f if line.strip()]
        self._cur = 0
        top[0].reshape(self.batch_size, 1, 1, self.num_classes)
        top[1].reshape(self.batch_size, self.num_classes, self.new_h // 8 + 1, self.new_w // 8 + 1)
        top[2].reshape(self.batch_size, 3, self.new_h, self.new_w)

    def reshape(self, bottom, top):
        pass

    def backward(self, top, propagate_down, bottom):
        pass

    def forward(self, bottom, top):
        for itt in range(self.batch_size):
            im, label, image_label = self.load_next_image()
            top[0].data[itt, ...] = image_label
            top[1].data[itt, ...] = label
            top[2].data[itt, ...] = im

    def load_next_image(self):
        from random import shuffle
        from .data import _imread_bgr, _imread_gray
        if self._cur == len(self.indexlist):
            self._cur = 0
            shuffle(self.indexlist)
        image_file_path, label_file_path = self.indexlist[self._cur]
        image = _imread_bgr(self.root_folder + image_file_path)
        label = _imread_gray(self.root_folder + label_file_path)
        self._cur += 1
        return self.preprocess(image, label)

    def preprocess(self, image, label):
        from scipy.ndimage import zoom
        image = zoom(np.asarray(image).astype('float32'),
                     (self.new_h / float(image.shape[0]), self.new_w / float(image.shape[1]), 1.0), order=1)
        image = image[:, :, [2, 1, 0]]
        image = image - self.mean
        image = image.transpose([2, 0, 1])
        h, w = label.shape
        cues = np.zeros((self.num_classes, h, w), dtype=np.uint8)
        ys, xs = np.nonzero(label != self.ignore_label)
        cues[label[ys, xs], ys, xs] = 1
        if self.is_mirror:
            flip = np.random.choice(2) * 2 - 1
            image = image[:, :, ::flip]
            cues = cues[:, :, ::flip]
        unique_inst = np.unique(label)
        unique_inst = unique_inst[unique_inst != self.ignore_label]
        image_label = np.zeros((1, 1, self.num_classes))
        image_label[0, 0, unique_inst] = 1
        return image, cues, image_label
