"""Deterministic synthetic inputs for the DSRG supervision path.

PASCAL VOC, the CAM/saliency cue pickle and the VGG weights are not available,
so tests and bench.py drive the path with seeded synthetic batches shaped like
the blobs of training/experiment/seed_mc/train-s.prototxt:

  images (B,3,321,321) f32  BGR, mean (104,117,123) subtracted   (train-s.prototxt:3-22)
  logits (B,21,41,41)  f32  stand-in for fc8-SEC                 (train-s.prototxt:744)
  labels (B,1,1,21)    f32  0/1, background always 1             (pylayers.py:378-379)
  cues   (B,21,41,41)  f32  0/1 localisation cues                (pylayers.py:381-382)
"""
import numpy as np

MEAN_PIXEL = np.array([104.0, 117.0, 123.0], dtype=np.float32)


def _smooth(rng, shape, sigma):
    from scipy.ndimage import gaussian_filter
    return gaussian_filter(rng.standard_normal(shape), sigma)


def make_images(rng, B, size=321, kind="smooth"):
    """(B,3,size,size) f32, integer-valued pixels minus the mean pixel."""
    from scipy.ndimage import gaussian_filter
    out = np.empty((B, 3, size, size), dtype=np.float32)
    for b in range(B):
        for c in range(3):
            if kind == "noise":
                px = rng.integers(0, 256, size=(size, size)).astype(np.float64)
            else:
                f = gaussian_filter(rng.random((size, size)), 6.0)
                f = (f - f.min()) / max(f.max() - f.min(), 1e-12)
                px = np.clip(np.rint(f * 255.0), 0, 255)
                if kind == "dark_corner":
                    px[: size // 3, : size // 3] = 0.0
            out[b, c] = px.astype(np.float32) - MEAN_PIXEL[c]
    return out


def make_logits(rng, B, C=21, H=41, W=41, gain=120.0, sigma=4.0):
    out = np.empty((B, C, H, W), dtype=np.float32)
    for b in range(B):
        for c in range(C):
            out[b, c] = (_smooth(rng, (H, W), sigma) * gain).astype(np.float32)
    return out


def make_labels_cues(rng, B, C=21, H=41, W=41):
    labels = np.zeros((B, 1, 1, C), dtype=np.float32)
    cues = np.zeros((B, C, H, W), dtype=np.float32)
    for b in range(B):
        nfg = int(rng.integers(1, 4))
        fg = rng.choice(np.arange(1, C), size=nfg, replace=False)
        labels[b, 0, 0, 0] = 1.0
        labels[b, 0, 0, fg] = 1.0
        for c in [0] + sorted(int(x) for x in fg):
            for _ in range(int(rng.integers(1, 5))):
                h, w = int(rng.integers(1, 6)), int(rng.integers(1, 6))
                y, x = int(rng.integers(0, H - h + 1)), int(rng.integers(0, W - w + 1))
                cues[b, c, y:y + h, x:x + w] = 1.0
        if b % 3 == 2:  # overlapping bg/fg pair -> pixels cued by two classes
            c = int(fg[0])
            y, x = int(rng.integers(0, H - 6)), int(rng.integers(0, W - 6))
            cues[b, 0, y:y + 4, x:x + 4] = 1.0
            cues[b, c, y + 2:y + 6, x + 2:x + 6] = 1.0
    return labels, cues


def make_batch(seed, B, C=21, H=41, W=41, size=321, image_kind="smooth"):
    """One synthetic train-s batch: dict(images, logits, labels, cues)."""
    rng = np.random.default_rng(seed)
    images = make_images(rng, B, size=size, kind=image_kind)
    logits = make_logits(rng, B, C, H, W)
    labels, cues = make_labels_cues(rng, B, C, H, W)
    return dict(images=images, logits=logits, labels=labels, cues=cues)
