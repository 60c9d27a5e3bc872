"""Host-side data path of the retrain stage: the drop-in for the reference module
`pylayers.layer` (pylayers/pylayers/layer.py) — list reader, pad / random-crop / mirror, BGR mean.

  ImageSegDataLayer  <-> layer.py:17-74     (Caffe Python-layer protocol; train-f.prototxt:3-14)
  BatchLoader        <-> layer.py:77-116
  SimpleTransformer  <-> layer.py:119-251

Pure data marshalling, done on the host like the reference; class, method and parameter names are the ones the prototxt's
`module: 'pylayers.layer'` and its `param_str` bind to.  The form is this repo's own: pad-then-crop is ONE placement of the
source window on a crop-sized canvas (`_Window`), never a padded copy of the whole image; the mirror is a boolean; the list
reader is a generator.  What has to match the reference exactly is kept and said where: which random draws are made, from
which generator, in which order and over which ranges (a training run seeded like the reference's sees the same crops).
Differences forced by the image: OpenCV is absent, so files are read with PIL (converted to OpenCV's BGR order); `param_str` is
parsed with ast.literal_eval, never eval (layer.py:30 uses eval).
"""
import ast
import random

import numpy as np

try:
    import caffe as _caffe
    _Base = _caffe.Layer
except ImportError:
    _Base = object

# layer.py:238-251: the keys a param_str may leave out
_DEFAULTS = (('crop_size', (505, 505)), ('mean', [128, 128, 128]), ('scale', 1.0), ('mirror', False), ('phase', 'Train'),
             ('ignore_label', 255))


class _Window(object):
    """Where a (crop_h, crop_w) crop sits on an image that is first extended to at least the crop size at its bottom / right
    edges (layer.py:156-166,200-225: copyMakeBorder + slicing).  `top`, `left` are the crop's offsets on the EXTENDED image;
    only the part of the crop that lies on the real image is ever copied."""

    def __init__(self, shape, crop, top=None, left=None):
        self.crop = crop
        self.ext = (max(shape[0], crop[0]), max(shape[1], crop[1]))         # size after the border extension
        # centred unless the caller drew the offsets ((ext - crop) // 2: the reference's py2 integer division)
        self.top = (self.ext[0] - crop[0]) // 2 if top is None else top
        self.left = (self.ext[1] - crop[1]) // 2 if left is None else left
        self.rows = max(0, min(shape[0] - self.top, crop[0]))                # rows / columns of the crop that hold image data
        self.cols = max(0, min(shape[1] - self.left, crop[1]))

    def slack(self):
        """largest admissible (top, left): the ranges the training phase draws its offsets from"""
        return self.ext[0] - self.crop[0], self.ext[1] - self.crop[1]

    def cut(self, a, fill):
        """the crop of `a` (H,W[,C]) as float32, border = fill"""
        out = np.full(tuple(self.crop) + a.shape[2:], fill, dtype=np.float32)
        out[:self.rows, :self.cols] = a[self.top:self.top + self.rows, self.left:self.left + self.cols]
        return out


class SimpleTransformer:
    """layer.py:119-251"""

    def __init__(self, params):
        self.check_params(params)
        self.crop_h, self.crop_w = params['crop_size']
        self.mean, self.scale = params['mean'], params['scale']
        self.is_mirror, self.phase, self.ignore_label = params['mirror'], params['phase'], params['ignore_label']

    def set_mean(self, mean):
        self.mean = mean

    def set_scale(self, scale):
        self.scale = scale

    def _centred(self, image):
        """mean-subtracted HWC image -> centre crop (zero border where the image is smaller), CHW"""
        return _Window(image.shape, (self.crop_h, self.crop_w)).cut(image, 0.0).transpose(2, 0, 1)

    def pre_test_image(self, image):
        """layer.py:150-169: an RGB image (PIL order) -> BGR, mean, centre crop, CHW; no scaling on this path"""
        bgr = np.asarray(image, np.float32)[:, :, ::-1]
        return self._centred(bgr - np.asarray(self.mean, np.float32))

    def preprocess(self, image, label=None):
        """layer.py:171-236: image is BGR (cv2.imread order) -> CHW float32 crop (and the label crop, float32).
        Random draws, as the reference makes them: `random.randint` for the row offset, then for the column offset, both over
        the slack of the EXTENDED image and only in phase 'Train'; then, only with `mirror`, ONE `np.random.choice(2)` whose
        value 0 means "mirror" (the reference's stride 2*0-1 = -1)."""
        x = (np.asarray(image, np.float32) - np.asarray(self.mean, np.float32)) * np.float32(self.scale)
        if label is None:
            return self._centred(x)
        label = np.asarray(label)
        crop = (self.crop_h, self.crop_w)
        win = _Window(label.shape, crop)
        if self.phase == 'Train':
            max_top, max_left = win.slack()
            top = random.randint(0, max_top)
            win = _Window(label.shape, crop, top, random.randint(0, max_left))
        img, lab = win.cut(x, 0.0).transpose(2, 0, 1), win.cut(label, self.ignore_label)
        mirrored = bool(self.is_mirror) and int(np.random.choice(2)) == 0
        return (img[:, :, ::-1], lab[:, ::-1]) if mirrored else (img, lab)

    @classmethod
    def check_params(cls, params):
        for key, value in _DEFAULTS:
            params.setdefault(key, value)


def _imread_bgr(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert('RGB'))[:, :, ::-1]


def _imread_gray(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert('L'))


class BatchLoader(object):
    """layer.py:77-116: `source` lists "image_path label_path" pairs relative to root_folder.  The list is walked in file order
    for the first epoch and reshuffled (`random.shuffle`, the generator the crop offsets also come from) before every later one."""

    def __init__(self, params):
        self.batch_size, self.root_folder, self.source = params['batch_size'], params['root_folder'], params['source']
        with open(self.source) as f:
            self.indexlist = [ln.split() for ln in f if ln.strip()]
        if not self.indexlist:
            raise ValueError("%s lists no image / label pair" % self.source)
        self.transformer = SimpleTransformer(params)
        self._epochs = self._walk()

    def _walk(self):
        while True:
            for image_path, label_path in list(self.indexlist):
                yield self.root_folder + image_path, self.root_folder + label_path
            random.shuffle(self.indexlist)

    def load_next_image(self):
        image_path, label_path = next(self._epochs)
        return self.transformer.preprocess(_imread_bgr(image_path), _imread_gray(label_path))


class ImageSegDataLayer(_Base):
    """layer.py:17-74: tops = [data (B,3,h,w), label (B,1,h,w)], filled image by image on the host"""

    top_names = ['data', 'label']

    def setup(self, bottom, top):
        params = ast.literal_eval(self.param_str)
        self.batch_loader = BatchLoader(params)                 # (its transformer fills in the defaults: crop_size is set below)
        self.batch_size, self.input_shape = params['batch_size'], tuple(params['crop_size'])
        for blob, channels in zip(top, (3, 1)):
            blob.reshape(self.batch_size, channels, *self.input_shape)

    def forward(self, bottom, top):
        data, label = top[0].data, top[1].data
        for n in range(self.batch_size):
            data[n], label[n, 0] = self.batch_loader.load_next_image()

    def reshape(self, bottom, top):
        pass                                                    # fixed crop size: shaped once in setup

    def backward(self, top, propagate_down, bottom):
        pass                                                    # a data layer has nothing to propagate
