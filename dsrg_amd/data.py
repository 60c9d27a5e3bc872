"""Host-side data path of the retrain stage: the drop-in for the reference module
`pylayers.layer` (pylayers/pylayers/layer.py) — list reader, pad / random-crop / mirror, BGR mean.

  ImageSegDataLayer  <-> layer.py:17-74     (Caffe Python-layer protocol; train-f.prototxt:3-14)
  BatchLoader        <-> layer.py:77-116
  SimpleTransformer  <-> layer.py:119-251

Pure data marshalling, done on the host like the reference.  Differences forced by the image:
OpenCV is absent, so files are read with PIL (converted to OpenCV's BGR order) and borders are
padded with numpy; `param_str` is parsed with ast.literal_eval, never eval (layer.py:30 uses eval).
"""
import ast
import random
from random import shuffle

import numpy as np

try:
    import caffe as _caffe
    _Base = _caffe.Layer
except ImportError:
    _Base = object


def _pad_bottom_right(a, pad_h, pad_w, value):
    """cv2.copyMakeBorder(a, 0, pad_h, 0, pad_w, BORDER_CONSTANT, value)"""
    if a.ndim == 3:
        out = np.empty((a.shape[0] + pad_h, a.shape[1] + pad_w, a.shape[2]), dtype=a.dtype)
        out[...] = np.asarray(value, dtype=a.dtype)
    else:
        out = np.full((a.shape[0] + pad_h, a.shape[1] + pad_w), value[0] if isinstance(value, tuple) else value, dtype=a.dtype)
    out[:a.shape[0], :a.shape[1]] = a
    return out


class SimpleTransformer:
    """layer.py:119-251"""

    def __init__(self, params):
        SimpleTransformer.check_params(params)
        self.mean = params['mean']
        self.is_mirror = params['mirror']
        self.crop_h, self.crop_w = params['crop_size']
        self.scale = params['scale']
        self.phase = params['phase']
        self.ignore_label = params['ignore_label']

    def set_mean(self, mean):
        self.mean = mean

    def set_scale(self, scale):
        self.scale = scale

    def _center_crop(self, img_pad):
        img_h, img_w = img_pad.shape[:2]
        h_off = (img_h - self.crop_h) // 2
        w_off = (img_w - self.crop_w) // 2
        return np.asarray(img_pad[h_off:h_off + self.crop_h, w_off:w_off + self.crop_w], np.float32)

    def _pad_image(self, image):
        img_h, img_w = image.shape[:2]
        pad_h = max(self.crop_h - img_h, 0)
        pad_w = max(self.crop_w - img_w, 0)
        return (_pad_bottom_right(image, pad_h, pad_w, (0.0, 0.0, 0.0)) if (pad_h > 0 or pad_w > 0) else image), pad_h, pad_w

    def pre_test_image(self, image):
        """layer.py:150-169: RGB -> BGR, mean, pad, centre crop, CHW"""
        image = np.asarray(image, np.float32)
        image = image[:, :, [2, 1, 0]]
        image = image - np.asarray(self.mean, np.float32)
        img_pad, _, _ = self._pad_image(image)
        return self._center_crop(img_pad).transpose((2, 0, 1))

    def preprocess(self, image, label=None):
        """layer.py:171-236: image is BGR (cv2.imread order); returns CHW float32 (and the label crop)"""
        image = np.asarray(image, np.float32).copy()
        image -= np.asarray(self.mean, np.float32)
        image *= self.scale
        if label is None:
            img_pad, _, _ = self._pad_image(image)
            return self._center_crop(img_pad).transpose((2, 0, 1))
        img_pad, pad_h, pad_w = self._pad_image(image)
        label_pad = _pad_bottom_right(np.asarray(label), pad_h, pad_w, (self.ignore_label,)) \
            if (pad_h > 0 or pad_w > 0) else np.asarray(label)
        img_h, img_w = label_pad.shape
        if self.phase == 'Train':
            h_off = random.randint(0, img_h - self.crop_h)
            w_off = random.randint(0, img_w - self.crop_w)
        else:
            h_off = (img_h - self.crop_h) // 2
            w_off = (img_w - self.crop_w) // 2
        image = np.asarray(img_pad[h_off:h_off + self.crop_h, w_off:w_off + self.crop_w], np.float32)
        label = np.asarray(label_pad[h_off:h_off + self.crop_h, w_off:w_off + self.crop_w], np.float32)
        image = image.transpose((2, 0, 1))
        if self.is_mirror:
            flip = np.random.choice(2) * 2 - 1
            image = image[:, :, ::flip]
            label = label[:, ::flip]
        return image, label

    @classmethod
    def check_params(cls, params):
        params.setdefault('crop_size', (505, 505))
        params.setdefault('mean', [128, 128, 128])
        params.setdefault('scale', 1.0)
        params.setdefault('mirror', False)
        params.setdefault('phase', 'Train')
        params.setdefault('ignore_label', 255)


def _imread_bgr(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert('RGB'))[:, :, ::-1]


def _imread_gray(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert('L'))


class BatchLoader(object):
    """layer.py:77-116: `source` lists "image_path label_path" pairs relative to root_folder"""

    def __init__(self, params):
        self.batch_size = params['batch_size']
        self.root_folder = params['root_folder']
        self.source = params['source']
        self.indexlist = [line.strip().split() for line in open(self.source) if line.strip()]
        self._cur = 0
        self.transformer = SimpleTransformer(params)

    def load_next_image(self):
        if self._cur == len(self.indexlist):
            self._cur = 0
            shuffle(self.indexlist)
        image_file_path, label_file_path = self.indexlist[self._cur]
        image = _imread_bgr(self.root_folder + image_file_path)
        label = _imread_gray(self.root_folder + label_file_path)
        self._cur += 1
        return self.transformer.preprocess(image, label)


class ImageSegDataLayer(_Base):
    """layer.py:17-74: tops = [data (B,3,h,w), label (B,1,h,w)]"""

    def setup(self, bottom, top):
        self.top_names = ['data', 'label']
        params = ast.literal_eval(self.param_str)
        SimpleTransformer.check_params(params)
        self.batch_size = params['batch_size']
        self.input_shape = params['crop_size']
        self.batch_loader = BatchLoader(params)
        top[0].reshape(self.batch_size, 3, self.input_shape[0], self.input_shape[1])
        top[1].reshape(self.batch_size, 1, self.input_shape[0], self.input_shape[1])

    def forward(self, bottom, top):
        for itt in range(self.batch_size):
            im, label = self.batch_loader.load_next_image()
            top[0].data[itt, ...] = im
            top[1].data[itt, ...] = label

    def reshape(self, bottom, top):
        pass

    def backward(self, top, propagate_down, bottom):
        pass
