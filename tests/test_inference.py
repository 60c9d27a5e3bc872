"""Test-time path: evaluation metrics on CPU; multi-scale inference + full-resolution
CRF on the GPU against a numpy/scipy restatement of training/tools/test-ms.py that uses the oracle CRF."""
import numpy as np
import pytest
import torch


def _reference_confusion(gt, pred, n):
    """evaluate.py:24-29,40-68 as written (loops)"""
    M = np.zeros((n, n))
    for g, p in zip(gt, pred):
        if not g == 255:
            M[g, p] += 1.0
    recall = sum(M[i, i] / np.sum(M[:, i]) for i in range(n)) / n
    acc = sum(M[i, i] / np.sum(M[i, :]) for i in range(n)) / n
    per = [M[i, i] / (np.sum(M[i, :]) + np.sum(M[:, i]) - M[i, i]) for i in range(n) if not M[i, i] == 0]
    return M, recall, acc, np.sum(per) / len(per)


def test_confusion_matrix_matches_reference_loops():
    from dsrg_amd.inference import ConfusionMatrix
    rng = np.random.default_rng(0)
    n = 21
    gt = rng.integers(0, n, size=5000)
    gt[rng.random(5000) < 0.1] = 255
    pred = np.where(rng.random(5000) < 0.6, np.where(gt == 255, 0, gt), rng.integers(0, n, size=5000))
    cm = ConfusionMatrix(n)
    cm.add(gt[:2000], pred[:2000])
    cm.add(gt[2000:], pred[2000:])
    M, recall, acc, miou = _reference_confusion(gt, pred, n)
    assert np.array_equal(cm.M, M)
    assert abs(cm.recall() - recall) < 1e-12 and abs(cm.accuracy() - acc) < 1e-12
    assert abs(cm.jaccard()[0] - miou) < 1e-12


class TinyNet(torch.nn.Module):
    """a deterministic stand-in for the deploy net: stride-8 feature map with 21 outputs"""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.w = torch.nn.Parameter(torch.randn(21, 3, 9, 9, generator=g) * 0.02)

    def forward(self, x):
        return torch.nn.functional.conv2d(x, self.w, stride=8, padding=4)


@pytest.mark.gpu
def test_predict_mask_ms_vs_reference_restatement():
    import scipy.ndimage as nd
    from dsrg_amd import inference as I, synthetic as S
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    H, W = 97, 131
    im = (S.make_images(rng, 1, size=max(H, W))[0, :, :H, :W] + S.MEAN_PIXEL[:, None, None]).transpose(1, 2, 0)
    im = np.ascontiguousarray(im[:, :, ::-1]).astype(np.uint8)            # an "RGB" uint8 image
    net = TinyNet().cuda().eval()
    got = I.predict_mask_ms(net, im, smooth=True)
    # test-ms.py:84-111 in numpy/scipy, network evaluated by the same TinyNet on the CPU in float64
    netc = TinyNet().double().eval()
    d1, d2 = float(H), float(W)
    scores_all = 0
    for size in [241, 321, 401]:
        x = nd.zoom(im.astype('float32'), (size / d1, size / d2, 1.0), order=1)[:, :, [2, 1, 0]] - np.array(I.MEAN_PIXEL)
        with torch.no_grad():
            sc = netc(torch.tensor(x.transpose(2, 0, 1)[None], dtype=torch.float64))[0].numpy().transpose(1, 2, 0)
        scores_all = scores_all + nd.zoom(sc, (d1 / sc.shape[0], d2 / sc.shape[1], 1.0), order=1)
    e = np.exp(scores_all - scores_all.max(2, keepdims=True))
    probs = e / e.sum(2, keepdims=True)
    probs[probs < 0.00001] = 0.00001
    q = O.CRF(im, np.log(probs), scale_factor=1.0)
    want = np.argmax(q, axis=2)
    agree = (got == want).mean()
    # the two pipelines differ upstream of the CRF (fp32 GPU convolution and torch's bilinear resampling against an fp64
    # convolution and scipy's zoom): a pixel may come out differently only where the reference's own decision is a near tie
    top2 = np.sort(q, axis=2)[:, :, -2:]
    margin = top2[:, :, 1] - top2[:, :, 0]
    bad = got != want
    print("multi-scale + CRF mask agreement with the reference restatement: %.5f; %d differing pixels, largest reference "
          "top-2 margin among them %.4f" % (agree, int(bad.sum()), float(margin[bad].max()) if bad.any() else 0.0))
    assert got.shape == (H, W) and agree > 0.999
    assert not bad.any() or margin[bad].max() < 1e-3
    assert (got[~bad] == want[~bad]).all() and (margin[~bad] >= 0).all()
    # pseudo-label generation restricted to the image labels (generate_train_gt.py:78-106)
    mask = I.predict_train_gt(net, im, labels=[3, 7], smooth=True)
    assert set(np.unique(mask)) <= {0, 3, 7}


@pytest.mark.parametrize("src,dst", [((241, 241), (366, 500)), ((321, 321), (366, 500)), ((401, 401), (366, 500)),
                                     ((375, 500), (241, 241)), ((281, 500), (401, 401)), ((41, 41), (321, 321)), ((7, 5), (7, 5))])
def test_zoom_is_scipy_ndimage_zoom_order_1(src, dst):
    """test-ms.py:77,96 resample with scipy.ndimage.zoom(order=1): the output grid maps onto the input with the
    (in - 1) / (out - 1) rule.  inference._zoom (align-corners bilinear interpolation) against scipy itself, both directions
    the pipeline uses (image -> network size, scores -> image size), to 1e-5 of the value range."""
    import scipy.ndimage as nd
    from dsrg_amd import inference as I
    rng = np.random.default_rng(src[0] * 1000 + dst[1])
    x = (rng.standard_normal((3,) + src) * 50).astype(np.float32)
    want = np.stack([nd.zoom(x[c], (dst[0] / float(src[0]), dst[1] / float(src[1])), order=1) for c in range(3)])
    assert want.shape == (3,) + dst
    got = I._zoom(torch.from_numpy(x)[None], dst[0], dst[1])[0].numpy()
    assert np.abs(got - want).max() <= 1e-5 * np.abs(x).max()
    if torch.cuda.is_available():
        got_gpu = I._zoom(torch.from_numpy(x)[None].cuda(), dst[0], dst[1])[0].cpu().numpy()
        assert np.abs(got_gpu - want).max() <= 1e-5 * np.abs(x).max()


def test_generate_m_rule():
    """evaluate.py:61-68 keeps ground truth < nclass (not only != 255)"""
    from dsrg_amd.inference import ConfusionMatrix
    from oracle import oracle as O
    rng = np.random.default_rng(2)
    gt = rng.integers(0, 21, size=3000).astype(np.uint8)
    gt[rng.random(3000) < 0.1] = 255
    gt[rng.random(3000) < 0.05] = 60
    pred = rng.integers(0, 21, size=3000).astype(np.uint8)
    cm = ConfusionMatrix(21)
    assert np.array_equal(cm.generateM((gt, pred)), O.confusion_matrix(gt, pred, 21, rule_lt=True))


@pytest.mark.gpu
def test_backbone_forward_replays_from_a_hip_graph():
    """bench.py --mode infer replays the VGG16-ASPP forward from a hipGraph (torch.cuda.CUDAGraph): every HIP kernel of the
    forward (direct convolutions, pooling, heads) takes torch's current stream, so it is capturable, and in eval mode the
    replayed scores equal the eager ones bit for bit — also after the input buffer has been overwritten in place"""
    from dsrg_amd.backbone import VGG16ASPP
    torch.manual_seed(3)
    dev = torch.device("cuda")
    net = VGG16ASPP().to(dev).to(memory_format=torch.channels_last).eval()
    x = torch.randn(1, 3, 97, 113, device=dev).contiguous(memory_format=torch.channels_last)

    from dsrg_amd.backbone import GraphedForward

    def fwd():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return net(x).contiguous()
    g = GraphedForward(net, x)
    for seed in (5, 6):
        torch.manual_seed(seed)
        x.copy_(torch.randn_like(x))
        static = g(x)
        assert static.dtype == torch.float32 and torch.equal(static, fwd())
    with pytest.raises(ValueError):
        GraphedForward(net.train(), x)


@pytest.mark.gpu
def test_graphed_forward_equals_the_eager_forward():
    """inference.GraphedForward (one captured HIP graph per input shape of the test-time loop) against the eager forward of the same
    VGG16-ASPP: bit-equal scores for several inputs and both shapes, weights updated in place are seen by the next replay, and
    predict_mask_ms gives the same mask either way"""
    from dsrg_amd import inference as I, synthetic as S
    from dsrg_amd.backbone import VGG16ASPP
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    net = VGG16ASPP().to(dev).to(memory_format=torch.channels_last).eval()
    fwd = I.GraphedForward(net)
    g = torch.Generator(device="cuda").manual_seed(1)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        for rep in range(3):
            for size in (241, 321):
                x = torch.randn(1, 3, size, size, device=dev, generator=g) * 40.0
                want = net(x).float().clone()
                got = fwd(x).float().clone()
                assert torch.equal(got, want), (rep, size)
        assert len(fwd._g) == 2
        with torch.no_grad():
            net.fc8[0].weight.mul_(0.5) if hasattr(net, "fc8") else [p.mul_(0.5) for p in list(net.parameters())[-2:]]
        x = torch.randn(1, 3, 241, 241, device=dev, generator=g) * 40.0
        assert torch.equal(fwd(x).float(), net(x).float())
        rng = np.random.default_rng(5)
        H, W = 120, 160
        im = (S.make_images(rng, 1, size=max(H, W))[0, :, :H, :W] + S.MEAN_PIXEL[:, None, None]).transpose(1, 2, 0)
        im = np.ascontiguousarray(im[:, :, ::-1]).clip(0, 255).astype(np.uint8)
        a = I.predict_mask_ms(net, im, smooth=True, device=dev)
        b = I.predict_mask_ms(net, im, smooth=True, device=dev, forward=fwd)
        assert np.array_equal(a, b)
    net.train()
    with pytest.raises(RuntimeError):
        I.GraphedForward(net)(torch.zeros(1, 3, 241, 241, device=dev))


@pytest.mark.gpu
def test_predict_masks_ms_many_equals_the_one_image_calls():
    """inference.predict_masks_ms_many (forwards of the next image while the CRFs before it are in flight) against predict_mask_ms image
    by image: same masks, in order, for images of two sizes, with and without graphed forwards and batched CRF calls"""
    from dsrg_amd import inference as I, synthetic as S
    rng = np.random.default_rng(9)
    ims = []
    for k, (H, W) in enumerate([(97, 131), (97, 131), (120, 90), (97, 131), (120, 90), (120, 90), (120, 90)]):
        im = (S.make_images(rng, 1, size=max(H, W), kind=["smooth", "noise", "dark_corner"][k % 3])[0, :, :H, :W] + S.MEAN_PIXEL[:, None, None])
        ims.append(np.ascontiguousarray(im.transpose(1, 2, 0)[:, :, ::-1]).clip(0, 255).astype(np.uint8))
    net = TinyNet().cuda().eval()
    want = [I.predict_mask_ms(net, im, smooth=True) for im in ims]
    for kw in (dict(in_flight=3), dict(in_flight=2, batch=2, forward=I.GraphedForward(net))):
        got = list(I.predict_masks_ms_many(net, ims, **kw))
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert a.dtype == np.int64 and np.array_equal(a, b)
