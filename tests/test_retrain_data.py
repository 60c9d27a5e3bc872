"""CPU: stage-2 (train-f) pieces and the data path."""
import os
import random

import numpy as np
import torch

from dsrg_amd import retrain as R
from dsrg_amd.data import SimpleTransformer, BatchLoader, ImageSegDataLayer


def test_interp_shrink_and_loss_semantics():
    lab = torch.arange(321 * 321, dtype=torch.float32).reshape(1, 1, 321, 321) % 21
    s = R.interp_shrink(lab, 8)
    assert s.shape == (1, 1, 41, 41) and torch.equal(s, lab[..., ::8, ::8])
    assert R.interp_shrink(torch.zeros(1, 1, 513, 513), 8).shape[-2:] == (65, 65)
    torch.manual_seed(0)
    logits = torch.randn(2, 21, 41, 41, dtype=torch.float64)
    label = torch.randint(0, 21, (2, 1, 41, 41))
    label[0, 0, :10] = 255
    loss = R.seg_softmax_loss(logits, label)
    lp = torch.log_softmax(logits.float(), 1)
    keep = label[:, 0] != 255
    manual = -(lp.gather(1, label.clamp(max=20)).squeeze(1)[keep]).sum() / keep.sum()     # VALID normalisation
    assert abs(loss.item() - manual.item()) < 1e-5
    assert 0.0 <= R.seg_accuracy(logits, label).item() <= 1.0
    assert abs(R.poly_lr(1e-3, 0, 20000) - 1e-3) < 1e-12 and R.poly_lr(1e-3, 20000, 20000) == 0.0
    assert abs(R.poly_lr(1e-3, 10000, 20000) - 1e-3 * 0.5 ** 0.9) < 1e-12


def test_resnet101_deeplab_shapes_and_retrain_step():
    net = R.ResNet101DeepLab(blocks=(1, 1, 2, 1))                       # same topology, fewer blocks (CPU test)
    x = torch.randn(1, 3, 129, 129)
    assert net(x).shape == (1, 21, 17, 17)                               # output stride 8: (129-1)/8+1
    tr = R.RetrainTrainer(torch.device("cpu"), backbone="resnet101", amp_dtype=None, net=net, max_iter=10)
    label = torch.randint(0, 21, (1, 1, 129, 129)).float()
    l0 = tr.step(x, label).item()
    for _ in range(3):
        l1 = tr.step(x, label).item()
    assert np.isfinite(l0) and np.isfinite(l1)
    full = R.ResNet101DeepLab()
    n = sum(p.numel() for p in full.parameters())
    assert 40e6 < n < 50e6                                               # ResNet-101 trunk + 4 ASPP heads


def test_simple_transformer_pad_crop_mirror():
    params = {'mean': (104.0, 117.0, 123.0), 'mirror': True, 'crop_size': (32, 40)}
    t = SimpleTransformer(params)
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(20, 50, 3)).astype(np.uint8)      # shorter than the crop, wider than it
    lab = rng.integers(0, 21, size=(20, 50)).astype(np.uint8)
    random.seed(1); np.random.seed(1)
    im, lb = t.preprocess(img, lab)
    assert im.shape == (3, 32, 40) and lb.shape == (32, 40) and im.dtype == np.float32
    # the padded rows carry the ignore label and a zero (mean-subtracted) image
    assert (lb[20:] == 255).all() and (im[:, 20:] == 0).all()
    # replay the random draws: same offsets, same flip
    random.seed(1); np.random.seed(1)
    h_off, w_off = random.randint(0, 32 - 32), random.randint(0, 50 - 40)
    flip = np.random.choice(2) * 2 - 1
    want = (img.astype(np.float32) - np.array(params['mean'], np.float32))[:, w_off:w_off + 40][:, ::flip]
    assert np.array_equal(im[:, :20], want.transpose(2, 0, 1))
    assert np.array_equal(lb[:20], lab[:, w_off:w_off + 40][:, ::flip].astype(np.float32))
    # test-phase path: centre crop, RGB -> BGR
    tt = SimpleTransformer({'mean': (1.0, 2.0, 3.0), 'crop_size': (16, 16), 'phase': 'Test'})
    x = tt.pre_test_image(img)
    assert x.shape == (3, 16, 16)
    assert np.array_equal(x[0], img[2:18, 17:33, 2].astype(np.float32) - 1.0)


def test_transformer_equals_pad_then_slice_on_random_shapes():
    """the crop window placed on a crop-sized canvas == the reference's sequence (extend the image at the bottom / right to the
    crop size, slice at the drawn offsets, mirror with stride -1 when np.random.choice(2) draws 0: layer.py:200-236), draws
    replayed from the same seeds"""
    rng = np.random.default_rng(5)
    for case in range(40):
        H, W = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        ch, cw = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        phase = 'Train' if case % 3 else 'Test'
        mirror = bool(case % 2)
        t = SimpleTransformer({'mean': (10.0, 20.0, 30.0), 'scale': 0.5, 'mirror': mirror, 'crop_size': (ch, cw), 'phase': phase,
                               'ignore_label': 254})
        img = rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8)
        lab = rng.integers(0, 21, size=(H, W)).astype(np.uint8)
        random.seed(case); np.random.seed(case)
        im, lb = t.preprocess(img, lab)
        st_py, st_np = random.getstate(), np.random.get_state()[2]
        random.seed(case); np.random.seed(case)
        ph, pw = max(ch - H, 0), max(cw - W, 0)
        x = np.pad((img.astype(np.float32) - np.array([10.0, 20.0, 30.0], np.float32)) * np.float32(0.5), ((0, ph), (0, pw), (0, 0)))
        y = np.pad(lab, ((0, ph), (0, pw)), constant_values=254)
        if phase == 'Train':
            ho = random.randint(0, y.shape[0] - ch); wo = random.randint(0, y.shape[1] - cw)
        else:
            ho, wo = (y.shape[0] - ch) // 2, (y.shape[1] - cw) // 2
        x, y = x[ho:ho + ch, wo:wo + cw].transpose(2, 0, 1), y[ho:ho + ch, wo:wo + cw].astype(np.float32)
        if mirror:
            st = np.random.choice(2) * 2 - 1
            x, y = x[:, :, ::st], y[:, ::st]
        assert im.dtype == np.float32 and lb.dtype == np.float32
        assert np.array_equal(im, x) and np.array_equal(lb, y), case
        assert random.getstate() == st_py and np.random.get_state()[2] == st_np      # same draws, no further ones
        # the label-less call: centre crop on the zero-extended image
        x0 = np.pad((img.astype(np.float32) - np.array([10.0, 20.0, 30.0], np.float32)) * np.float32(0.5), ((0, ph), (0, pw), (0, 0)))
        ho, wo = (x0.shape[0] - ch) // 2, (x0.shape[1] - cw) // 2
        assert np.array_equal(t.preprocess(img), x0[ho:ho + ch, wo:wo + cw].transpose(2, 0, 1))


def test_batch_loader_walks_in_file_order_then_reshuffles(tmp_path):
    from PIL import Image
    root = str(tmp_path) + "/"
    for k in range(4):
        Image.fromarray(np.full((8, 8, 3), 10 * k, np.uint8)).save(root + "i%d.png" % k)
        Image.fromarray(np.full((8, 8), k, np.uint8)).save(root + "l%d.png" % k)
    src = root + "list.txt"
    open(src, "w").write("".join("i%d.png l%d.png\n" % (k, k) for k in range(4)) + "\n")
    random.seed(11)
    ld = BatchLoader({'batch_size': 2, 'root_folder': root, 'source': src, 'mean': (0.0, 0.0, 0.0), 'crop_size': (8, 8), 'phase': 'Test'})
    seen = [int(ld.load_next_image()[1][0, 0]) for _ in range(12)]
    assert seen[:4] == [0, 1, 2, 3]                                       # first epoch: file order (layer.py:99-103)
    random.seed(11)
    order = [0, 1, 2, 3]
    random.shuffle(order); assert seen[4:8] == order                      # reshuffled at the epoch boundary by random.shuffle ...
    random.shuffle(order); assert seen[8:12] == order                     # ... in place, epoch after epoch


def test_image_seg_data_layer_protocol(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(1)
    root = str(tmp_path) + "/"
    lines = []
    for k in range(3):
        Image.fromarray(rng.integers(0, 256, size=(30, 36, 3)).astype(np.uint8)).save(root + "im%d.png" % k)
        Image.fromarray(rng.integers(0, 21, size=(30, 36)).astype(np.uint8)).save(root + "lb%d.png" % k)
        lines.append("im%d.png lb%d.png" % (k, k))
    src = os.path.join(root, "train.txt")
    open(src, "w").write("\n".join(lines) + "\n")

    class Blob(object):
        def __init__(self):
            self.data = np.zeros((0,), np.float32)

        def reshape(self, *s):
            self.data = np.zeros(s, np.float32)
    layer = ImageSegDataLayer()
    layer.param_str = "{'batch_size': 4, 'root_folder': %r, 'mean': (104.0, 117.0, 123.0), 'source': %r, " \
                      "'mirror': True, 'crop_size': (24, 24)}" % (root, src)
    tops = [Blob(), Blob()]
    layer.setup([], tops)
    layer.forward([], tops)                                              # 4 images from a 3-image list: wraps + reshuffles
    assert tops[0].data.shape == (4, 3, 24, 24) and tops[1].data.shape == (4, 1, 24, 24)
    assert tops[1].data.max() <= 20 and np.isfinite(tops[0].data).all()
    import pylayers.layer as PL
    assert PL.ImageSegDataLayer is ImageSegDataLayer and PL.BatchLoader is BatchLoader
