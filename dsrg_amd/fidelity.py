"""bf16 step against float32 step: gradient fidelity of the backbone's backward chain (measurement tooling shared by
tools/grad_fidelity.py, bench.py's `legs` and tests/test_gpu_trainer.py; not on the hot path).

VGG16-ASPP with an ImageNet-scale initialisation (Kaiming fan-in for every ReLU layer, so activations stay O(input) through the
net as they do under vgg16_20M_mc.caffemodel, run.sh:5; fc8-SEC N(0, 0.01) as the prototxt), the bench's synthetic images,
Dropout off (the bf16 leg draws its masks in the convolution epilogues, the float32 leg from torch: different masks would hide
the arithmetic).  The score gradient is the train-s loss gradient taken ONCE at the float32 scores and fed to every leg, so what
is compared is the backbone's backward chain alone.  Legs, each against the float32 backbone (the reference's Caffe arithmetic):

  bf16        the shipped route (direct + implicit-GEMM kernels, fp32 master weights, fp32 heads)
  stock       torch's own bf16 autocast (F.conv2d / max_pool2d through MIOpen) — what "bf16 mixed precision" means elsewhere
  f32+act     float32 arithmetic, every activation rounded to bf16 where the bf16 route stores one
  f32+grad    float32 arithmetic, every activation gradient rounded to bf16 where the bf16 route stores one
  f32+both    both roundings (the bf16 route's storage precision with float32 products)
  act,amax2   f32+act, but pool1-3 (stride 2) pick each window's maximum on the unrounded convolution output
  act,amax12  ... and pool4 / pool5 (stride 1) too
  f32,w16     float32 arithmetic and storage throughout, only the WEIGHTS rounded to bf16 once (a 2^-9 relative perturbation of
              the point at which the float32 gradient is taken) — how far the float32 gradient itself moves under a perturbation
              of the size of bf16's rounding: the yardstick for the legs above
"""
import torch
import torch.nn.functional as F

from . import backbone

CL = torch.channels_last
LEGS = ("bf16", "stock", "f32plain", "f32+act", "f32+grad", "f32+both", "act,amax2", "act,amax12", "f32,w16")


def kaiming_(net, seed=5):
    """Kaiming fan-in for every ReLU layer; conv1_1 additionally divided by 64 (~ the standard deviation of a mean-subtracted
    8-bit image), so that activations are O(1) from conv1_1 on, as they are in a net trained on such images; fc8-SEC N(0, 0.01)"""
    g = torch.Generator().manual_seed(seed)
    for name, m in net.named_modules():
        if isinstance(m, torch.nn.Conv2d):
            fan_in = m.in_channels * m.kernel_size[0] * m.kernel_size[1]
            std = 0.01 if m.out_channels == 21 else (2.0 / fan_in) ** 0.5 / (64.0 if m.in_channels == 3 else 1.0)
            with torch.no_grad():
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * std)
                m.bias.zero_()
    return net


class _Round(torch.autograd.Function):
    """storage rounding of the bf16 route imitated in float32: value and / or gradient through bf16"""
    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return x.bfloat16().float() if fwd else x

    @staticmethod
    def backward(ctx, g):
        return (g.bfloat16().float() if ctx.bwd else g), None, None


def plain_forward(net, x, rnd=(False, False), argmax32=()):
    """the prototxt's layer sequence on torch ops; rnd = (round activations, round activation gradients) at every stored blob;
    argmax32: strides of the max pools that pick their window maximum on the UNROUNDED convolution output (the value stored is
    the same — rounding is monotone — but ties between bf16-equal neighbours are decided as float32 decides them)"""
    r = lambda t: _Round.apply(t, rnd[0], rnd[1]) if (rnd[0] or rnd[1]) else t          # noqa: E731

    def conv(m, h, training, raw_out=False):
        h = F.conv2d(h, m.weight, m.bias, 1, m.padding, m.dilation)
        if getattr(m, "fuse_relu", False):
            h = F.relu(h)
        if getattr(m, "fuse_pool", None) is not None:
            h = h if 2 in argmax32 else r(h)
            return r(F.max_pool2d(h, 3, 2, 1, ceil_mode=True))
        return h if raw_out else r(h)
    h = r(x)
    feats = [m for m in net.features if isinstance(m, (backbone.GemmConv2d, backbone.MaxPool3x3, backbone.AvgPool3x3))]
    for i, m in enumerate(feats):
        if isinstance(m, backbone.GemmConv2d):
            h = conv(m, h, net.training, raw_out=1 in argmax32 and isinstance(feats[i + 1], backbone.MaxPool3x3))
        elif isinstance(m, backbone.MaxPool3x3):
            h = r(F.max_pool2d(h, 3, 1, 1))
        elif isinstance(m, backbone.AvgPool3x3):
            h = r(F.avg_pool2d(h.contiguous(), 3, 1, 1))       # (torch's channels_last avg_pool backward is wrong on this build)
    out = None
    for br in net.branches:
        t = h
        for m in br:
            if isinstance(m, backbone.GemmConv2d):
                if m.out_channels == 21:
                    with torch.autocast("cuda", enabled=False):
                        s = F.conv2d(t.float(), m.weight, m.bias)
                    out = s if out is None else out + s
                else:
                    t = conv(m, t, net.training)
    return out


def gradient_fidelity(B=2, legs=("bf16",), seed=77, own_loss=False, log=None):
    """-> dict(ref_norm={param: |g| fp32}, cos={leg: {param: cosine}}, rel={leg: {param: relative distance}},
    cos_all={leg: cosine of the whole gradient}, score_rel={leg: relative distance of the scores}).  Needs a GPU."""
    from . import synthetic as S
    from .ops import dsrg_supervision_loss
    dev = torch.device("cuda", torch.cuda.current_device())
    b = S.make_batch(seed, B)
    images, labels, cues = (torch.from_numpy(b[k]).to(dev) for k in ("images", "labels", "cues"))
    x = images.contiguous(memory_format=CL)
    net = kaiming_(backbone.VGG16ASPP(dropout=0.0)).to(dev).to(memory_format=CL)
    say = log or (lambda *a: None)

    def run(tag, gout):
        net.zero_grad(set_to_none=True)
        saved = None
        if tag == "f32,w16":
            saved = [p.detach().clone() for p in net.parameters()]
            with torch.no_grad():
                for p in net.parameters():
                    p.copy_(p.bfloat16().float())
        try:
            if tag in ("fp32", "f32,w16"):
                y = net(x)
            elif tag == "bf16":
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = net(x)
            elif tag == "stock":
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = plain_forward(net, x)
            else:
                y = plain_forward(net, x, {"f32+act": (True, False), "f32+grad": (False, True), "f32+both": (True, True),
                                           "f32plain": (False, False), "act,amax2": (True, False), "act,amax12": (True, False)}[tag],
                                  argmax32={"act,amax2": (2,), "act,amax12": (1, 2)}.get(tag, ()))
            y = y.float().contiguous()
            if gout is None or own_loss:
                yl = y.detach().requires_grad_(True)
                total, losses = dsrg_supervision_loss(yl, images, labels, cues)
                total.backward()
                g = yl.grad.detach()
                say("# %-9s losses %s  |score grad| %.4e" % (tag, [round(float(v), 5) for v in losses], float(g.norm())))
            else:
                g = gout
            y.backward(g)
        finally:
            if saved is not None:
                with torch.no_grad():
                    for p, q in zip(net.parameters(), saved):
                        p.copy_(q)
        return y.detach(), g, {n: p.grad.detach().float().clone() for n, p in net.named_parameters()}

    y32, gout, ref = run("fp32", None)
    say("# scores: rms %.3f  max %.3f; batch %d" % (float(y32.pow(2).mean().sqrt()), float(y32.abs().max()), B))
    cosf = lambda a, c: float((a * c).sum() / (a.norm() * c.norm()).clamp_min(1e-30))     # noqa: E731
    relf = lambda a, c: float((a - c).norm() / c.norm().clamp_min(1e-30))                 # noqa: E731
    out = {"ref_norm": {n: float(v.norm()) for n, v in ref.items()}, "cos": {}, "rel": {}, "cos_all": {}, "score_rel": {}}
    flat_ref = torch.cat([ref[n].flatten() for n in ref])
    for tag in legs:
        y, _, gr = run(tag, gout)
        out["score_rel"][tag] = relf(y, y32)
        out["cos"][tag] = {n: cosf(gr[n], ref[n]) for n in ref}
        out["rel"][tag] = {n: relf(gr[n], ref[n]) for n in ref}
        out["cos_all"][tag] = cosf(torch.cat([gr[n].flatten() for n in ref]), flat_ref)
        say("# %-9s scores vs float32: rel %.3e" % (tag, out["score_rel"][tag]))
    del net
    return out
