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


def test_caffe_sgd_bumps_version_counters_and_packs_are_kept_only_while_their_owner_lives():
    """the contract the kept weight packs rest on (dsrg_amd/ops.py): (a) torch._fused_sgd_ itself leaves a parameter's version
    counter alone — the reason a counter cannot be the only staleness test — while every CaffeSGD.step() bumps it; (b) packs are
    kept only for parameters claimed by a live owner, and a new claim forgets what an earlier one left"""
    import gc
    from dsrg_amd import ops
    p = torch.nn.Parameter(torch.randn(8))
    g, b = torch.randn(8), torch.zeros(8)
    v0 = p._version
    try:
        with torch.no_grad():
            torch._fused_sgd_([p], [g], [b], weight_decay=0.0, momentum=0.9, lr=0.1, dampening=0.0, nesterov=False, maximize=False,
                              is_first_step=False)
        fused_bumps = p._version > v0                  # False on torch 2.10; if a later torch bumps it, all the better
    except (RuntimeError, NotImplementedError):
        fused_bumps = None
    assert fused_bumps in (False, True, None)
    opt = CaffeSGD([dict(params=[p], lr_mult=1.0, decay_mult=1.0)], base_lr=0.1)
    p.grad = g.clone()
    v1 = p._version
    opt.step()
    assert p._version > v1

    class Owner(object):
        pass
    w = torch.nn.Parameter(torch.randn(64, 64, 3, 3))
    assert not ops._packs_claimed(w)
    o1 = Owner()
    ops.keep_weight_packs([w], o1)
    assert ops._packs_claimed(w)
    e = ops._WeightPacks()
    e.ref, e.version, e.fwd, e.dg, e.plain = ops._weakref.ref(w), w._version, torch.zeros(1), None, 0
    ops._weight_packs[w.data_ptr()] = e
    assert ops._packs_entry(w, 0) is e and ops._packs_entry(w, 1) is None     # another layout: the entry goes
    ops._weight_packs[w.data_ptr()] = e
    del o1
    gc.collect()
    assert not ops._packs_claimed(w)
    ops.keep_weight_packs([w], Owner())                # the owner dies at once; the claim forgot the old entry
    assert w.data_ptr() not in ops._weight_packs and not ops._packs_claimed(w)
