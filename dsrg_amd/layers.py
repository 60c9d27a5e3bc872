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


def _dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device="cuda", dtype=dtype)


class SoftmaxLayer(_Base):
    """pylayers.py:23-51"""

    def setup(self, bottom, top):
        if len(bottom) != 1:
            raise Exception("Need two inputs to compute distance.")

    def reshape(self, bottom, top):
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
