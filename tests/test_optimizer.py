"""CPU: CaffeSGD (kept in the B = V / lr form of torch._fused_sgd_) against the literal Caffe SGDSolver update
V <- m V + lr (g + wd W); W <- W - V of solver-s.prototxt:5-14, across learning-rate steps and per-group multipliers."""
import torch

from dsrg_amd.trainer import CaffeSGD


def test_caffe_sgd_matches_literal_update_across_lr_steps():
    torch.manual_seed(0)
    ps = [torch.randn(5, 3).requires_grad_(True), torch.randn(7).requires_grad_(True)]
    mults = [(1.0, 1.0), (2.0, 0.0)]                       # weights (lr 1, decay 1), biases (lr 2, decay 0)
    groups = [dict(params=[p], lr_mult=m[0], decay_mult=m[1]) for p, m in zip(ps, mults)]
    opt = CaffeSGD(groups, base_lr=0.1, momentum=0.9, weight_decay=5e-4, gamma=0.33, stepsize=3)
    W = [p.detach().double().clone() for p in ps]
    V = [torch.zeros_like(w) for w in W]
    for it in range(11):
        gs = [torch.randn_like(p) for p in ps]
        for p, g in zip(ps, gs):
            p.grad = g.clone()
        lr = 0.1 * 0.33 ** (it // 3)
        assert abs(opt.lr() - lr) < 1e-15
        for k, (lm, dm) in enumerate(mults):
            V[k] = 0.9 * V[k] + lr * lm * (gs[k].double() + 5e-4 * dm * W[k])
            W[k] = W[k] - V[k]
        opt.step()
    for p, w in zip(ps, W):
        assert (p.detach().double() - w).abs().max() < 5e-6


def test_caffe_sgd_skips_parameters_without_gradient():
    a, b = torch.ones(3, requires_grad=True), torch.ones(3, requires_grad=True)
    opt = CaffeSGD([dict(params=[a, b], lr_mult=1.0, decay_mult=0.0)], base_lr=0.5)
    a.grad = torch.ones(3)
    opt.step()
    assert torch.allclose(a.detach(), torch.full((3,), 0.5)) and torch.equal(b.detach(), torch.ones(3))
