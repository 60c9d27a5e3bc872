"""CPU: the closed-form pointwise layers of the oracle against torch autograd in float64
(Theano is absent, so the Theano graphs of pylayers.py:30-41,126-142,160-168 are pinned by
re-deriving their gradients with an independent autodiff)."""
import numpy as np
import torch

from dsrg_amd import synthetic as S
from oracle import oracle as O


def _batch(seed=0, B=3, C=21, H=9, W=11):
    rng = np.random.default_rng(seed)
    logits = S.make_logits(rng, B, C, H, W, gain=30.0, sigma=2.0)
    labels, cues = S.make_labels_cues(rng, B, C, H, W)
    return logits, labels, cues


def _softmax_t(x):
    s = torch.softmax(x, dim=1) + 1e-4
    return s / s.sum(1, keepdim=True)


def test_softmax_forward_backward():
    logits, _, _ = _batch(0)
    p = O.softmax_forward(logits)
    xt = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
    pt = _softmax_t(xt)
    assert np.abs(p - pt.detach().numpy()).max() < 1e-6
    g = np.random.default_rng(1).standard_normal(logits.shape).astype(np.float32)
    (pt * torch.tensor(g, dtype=torch.float64)).sum().backward()
    dx = O.softmax_backward(logits, g)
    assert np.abs(dx - xt.grad.numpy()).max() < 1e-6


def test_balanced_seed_loss():
    logits, _, cues = _batch(2)
    p = O.softmax_forward(logits)
    cues[1] = 0.0                                   # an image without any seed: max(count,1e-4) guard
    loss, grad = O.seed_loss(p, cues)
    pt = torch.tensor(p, dtype=torch.float64, requires_grad=True)
    St = torch.tensor(cues, dtype=torch.float64)
    cb = St[:, 0].sum((1, 2), keepdim=True)
    cf = St[:, 1:].sum((1, 2, 3), keepdim=True)
    l1 = -((St[:, 0] * torch.log(pt[:, 0])).sum((1, 2), keepdim=True) / torch.clamp(cb, min=1e-4)).mean()
    l2 = -((St[:, 1:] * torch.log(pt[:, 1:])).sum((1, 2, 3), keepdim=True) / torch.clamp(cf, min=1e-4)).mean()
    lt = l1 + l2
    lt.backward()
    assert abs(loss - lt.item()) < 1e-9 * max(1, abs(lt.item()))
    assert np.abs(grad - pt.grad.numpy()).max() < 1e-5 * max(1.0, np.abs(pt.grad.numpy()).max())


def test_constrain_loss():
    logits, _, _ = _batch(3)
    p = O.softmax_forward(logits)
    rng = np.random.default_rng(4)
    q = O.softmax_forward((logits + rng.standard_normal(logits.shape) * 8).astype(np.float32))
    lq = np.log(q).astype(np.float32)
    loss, gp, gq = O.constrain_loss(p, lq)
    pt = torch.tensor(p, dtype=torch.float64, requires_grad=True)
    lt = torch.tensor(lq, dtype=torch.float64, requires_grad=True)
    qt = torch.exp(lt)
    L = (qt * torch.log(torch.clamp(qt / pt, 0.05, 20.0))).sum(1).mean()
    L.backward()
    assert abs(loss - L.item()) < 1e-9
    r = np.exp(lq.astype(np.float64)) / p
    assert ((r < 0.05) | (r > 20)).any(), "clip never fires; test is vacuous"
    assert np.abs(gp - pt.grad.numpy()).max() < 1e-7
    assert np.abs(gq - lt.grad.numpy()).max() < 1e-7


def test_plain_seed_loss():
    """SeedLossLayer (pylayers.py:94-118) against its Theano graph restated in torch float64"""
    logits, _, cues = _batch(5)
    p = O.softmax_forward(logits)
    loss, grad = O.seed_loss_plain(p, cues)
    pt = torch.tensor(p, dtype=torch.float64, requires_grad=True)
    St = torch.tensor(cues, dtype=torch.float64)
    count = St.sum((1, 2, 3), keepdim=True)
    lt = -((St * torch.log(pt)).sum((1, 2, 3), keepdim=True) / count).mean()
    lt.backward()
    assert abs(loss - lt.item()) < 1e-9 * max(1, abs(lt.item()))
    assert np.abs(grad - pt.grad.numpy()).max() < 1e-5 * max(1.0, np.abs(pt.grad.numpy()).max())


def _expand_loss_torch(p, stat, q_fg=0.996, q_bg=0.999):
    """the Theano graph of ExpandLossLayer.setup (pylayers.py:190-222), op for op, in torch float64"""
    B, C, H, W = p.shape
    pt = torch.tensor(p, dtype=torch.float64, requires_grad=True)
    st = torch.tensor(stat, dtype=torch.float64)[:, :, :, 1:]
    probs_bg, probs = pt[:, 0], pt[:, 1:]
    probs_max = probs.amax(3).amax(2) if False else probs.reshape(B, C - 1, -1).max(2).values
    n = H * W
    w_fg = torch.tensor(np.array([q_fg ** i for i in range(n - 1, -1, -1)]))[None, None, :]
    w_bg = torch.tensor(np.array([q_bg ** i for i in range(n - 1, -1, -1)]))[None, :]
    probs_mean = ((probs.reshape(B, C - 1, n).sort(2).values * w_fg) / w_fg.sum()).sum(2)
    bg_mean = ((probs_bg.reshape(B, n).sort(1).values * w_bg) / w_bg.sum()).sum(1)
    s2 = (st[:, 0, 0, :] > 0.5).double()
    l1 = -((s2 * torch.log(probs_mean) / s2.sum(1, keepdim=True)).sum(1)).mean()
    l2 = -(((1 - s2) * torch.log(1 - probs_max) / (1 - s2).sum(1, keepdim=True)).sum(1)).mean()
    l3 = -torch.log(bg_mean).mean()
    loss = l1 + l2 + l3
    loss.backward()
    return loss.item(), pt.grad.numpy()


def test_expand_loss():
    """ExpandLossLayer (pylayers.py:183-233): sort-weighted pooling, max pooling of absent classes, background term"""
    rng = np.random.default_rng(7)
    B, C, H, W = 3, 21, 9, 11
    logits = S.make_logits(rng, B, C, H, W, gain=6.0, sigma=2.0)
    p = O.softmax_forward(logits)
    stat = np.zeros((B, 1, 1, C), np.float32)
    stat[0, 0, 0, [0, 3, 7]] = 1
    stat[1, 0, 0, [0, 20]] = 1
    stat[2, 0, 0, [1, 2, 3, 4]] = 1                      # stat[:, 0] (background) is never read
    loss, grad = O.expand_loss(p, stat)
    lt, gt = _expand_loss_torch(p, stat)
    assert abs(loss - lt) < 1e-9 * max(1, abs(lt))
    assert np.abs(grad - gt).max() < 1e-5 * max(1.0, np.abs(gt).max())
    # the largest value of a present plane carries weight 1/Z, the smallest q^(n-1)/Z
    b, c = 0, 3
    k_hi, k_lo = p[b, c].argmax(), p[b, c].argmin()
    ratio = grad[b, c].ravel()[k_lo] / grad[b, c].ravel()[k_hi]
    assert abs(ratio - 0.996 ** (H * W - 1)) < 1e-4


def test_confusion_matrix_oracle():
    """evaluate.py:25-30,61-68 against the literal loops"""
    rng = np.random.default_rng(8)
    n = 21
    gt = rng.integers(0, n, size=4000).astype(np.uint8)
    gt[rng.random(4000) < 0.1] = 255
    gt[rng.random(4000) < 0.02] = 100                     # a label >= nclass that is not 255: the two rules differ
    pred = rng.integers(0, n, size=4000).astype(np.uint8)
    for rule_lt in (False, True):
        M = np.zeros((n, n))
        for g, q in zip(gt, pred):
            if (g < n) if rule_lt else (g != 255 and g < n):
                M[g, q] += 1.0
        keep = gt != 100 if not rule_lt else np.ones_like(gt, bool)
        assert np.array_equal(O.confusion_matrix(gt[keep], pred[keep], n, rule_lt), M)


def test_baseline_worker_script_runs():
    """bench.py's image-parallel CPU baseline launches oracle/baseline_worker.py: it must print '<count> <seconds>'"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "oracle", "baseline_worker.py"), "1", "0.3"],
                         capture_output=True, timeout=120, check=True).stdout.decode().split()
    assert int(out[0]) >= 1 and float(out[1]) >= 0.3
