"""Stage 2 of the seed_mc recipe: retraining on the CRF-refined pseudo-labels,
training/experiment/seed_mc/train-f.prototxt + solver-f.prototxt, and the DeepLab-v2 ResNet-101
backbone of BASELINE.json configs[4] (no counterpart in the reference: new work).

  interp_shrink        <-> DeepLab `Interp` layer, shrink_factor 8      (train-f.prototxt:721-731)
  seg_softmax_loss     <-> SoftmaxWithLoss, ignore_label 255            (train-f.prototxt:732-744)
  seg_accuracy         <-> SegAccuracy, ignore_label 255                (train-f.prototxt:745-755)
  poly_lr              <-> lr_policy "poly", power 0.9                  (solver-f.prototxt:5-7)

There is no SRG/CRF inside this step; it is plain PyTorch-ROCm plumbing around the same backbone.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

import os as _os

from .backbone import VGG16ASPP, GemmConv2d, _ConvFn
from .trainer import CaffeSGD

_IGEMM_BN = _os.environ.get("DSRG_RESNET_IGEMM", "1") == "1"      # tools: A/B against round 5's im2col + library-GEMM bottlenecks
_MERGED = _os.environ.get("DSRG_RESNET_MERGED_BWD", "1") == "1"   # tools / tests: 0 = data and weight gradient of a bottleneck convolution as two launches
_FUSE_RES = _os.environ.get("DSRG_RESNET_FUSE_RES", "1") == "1"   # tools / tests: 0 = the shortcut's add + ReLU and its backward as passes of their own


def interp_shrink(label, factor=8):
    """(B,1,H,W) -> (B,1,(H-1)/factor+1,(W-1)/factor+1).  DeepLab's Interp resamples with the
    align-corners mapping; at an exact shrink factor it lands on label[::factor, ::factor]."""
    H, W = label.shape[-2:]
    if (H - 1) % factor == 0 and (W - 1) % factor == 0:
        return label[..., ::factor, ::factor]
    h, w = (H - 1) // factor + 1, (W - 1) // factor + 1
    return F.interpolate(label.float(), size=(h, w), mode="bilinear", align_corners=True)


def seg_softmax_loss(logits, label, ignore_label=255):
    """mean over the non-ignored pixels of -log softmax(logits)[label] (Caffe VALID normalisation)"""
    return F.cross_entropy(logits.float(), label.reshape(label.shape[0], *label.shape[-2:]).long(),
                           ignore_index=ignore_label, reduction="mean")


def seg_accuracy(logits, label, ignore_label=255):
    lab = label.reshape(label.shape[0], *label.shape[-2:]).long()
    keep = lab != ignore_label
    return ((logits.argmax(1) == lab) & keep).sum().float() / keep.sum().clamp(min=1).float()


def poly_lr(base_lr, it, max_iter, power=0.9):
    return base_lr * (1.0 - float(it) / float(max_iter)) ** power


class _FrozenBN(nn.Module):
    """BatchNorm with fixed statistics (DeepLab-v2 trains ResNet-101 with use_global_stats).  train_affine=False (the default of
    ResNet101DeepLab): gamma and beta are fixed too — the layer is a constant per-channel affine map, which the implicit-GEMM
    route folds into the convolution in front of it (scale into the packed kernel, shift as the epilogue's bias)"""

    def __init__(self, c, train_affine=True):
        super().__init__()
        self.weight, self.bias = nn.Parameter(torch.ones(c), requires_grad=train_affine), nn.Parameter(torch.zeros(c), requires_grad=train_affine)
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self._folded = None                       # (versions, scale, shift) of the frozen map

    def affine(self):
        scale = self.weight * torch.rsqrt(self.running_var + 1e-5)
        return scale, self.bias - self.running_mean * scale

    def frozen_affine(self):
        """(scale, shift) float32, computed once per value of the four tensors (their version counters)"""
        key = (self.weight._version, self.bias._version, self.running_mean._version, self.running_var._version, self.weight.device)
        if self._folded is None or self._folded[0] != key:
            with torch.no_grad():
                scale, shift = self.affine()
                self._folded = (key, scale.float().contiguous(), shift.float().contiguous())
        return self._folded[1], self._folded[2]

    def forward(self, x):
        scale, shift = self.affine()
        return x * scale.view(1, -1, 1, 1).to(x.dtype) + shift.view(1, -1, 1, 1).to(x.dtype)


class _FoldedIgemmFn(torch.autograd.Function):
    """conv (1x1 or dilated 3x3, stride 1, 'same') -> constant per-channel affine (-> ReLU) of a ResNet bottleneck on the
    implicit-GEMM kernels of the VGG path (ops.conv_igemm): the scale rides in the packed bf16 kernel (packed from w * scale, both
    the forward and the flipped / transposed data-gradient form in one pass), the shift is the epilogue's bias.  Backward: ReLU
    mask, data gradient by the same kernel, weight gradient by the implicit-GEMM weight-gradient kernel where its tiling takes the
    shape (both sides multiples of 256 channels) and by the library otherwise; d/dw = scale * d/d(w * scale).
    x: bf16 channels_last; w: the float32 master parameter (channels_last); scale, shift: float32 (cout), no gradient."""

    @staticmethod
    def forward(ctx, x, w, scale, shift, dil, relu, link_in=None, link_out=None, res=None, res_link=None):
        """link_in / link_out (backbone._GradLink or None): x is the ReLU output of the node in front and feeds nothing but this
        node (its ReLU backward then rides in this node's data-gradient store) / this node's output is such an x for the next.
        res (bf16 channels_last, the output's shape): the block's shortcut, added in the store BEFORE the ReLU (the block's last
        convolution: y = relu(bf16(conv + shift) + res), ops.conv_igemm_residual).
        res_link (_ResLink): the block's last node leaves the masked gradient of the block output there instead of returning it for
        `res` when the block's first node said it would add it in its own data-gradient store (res is the block input itself)."""
        from .ops import conv_igemm, conv_igemm_residual, pack_conv_weight_pair
        k = w.shape[2]
        x = x if x.dtype == torch.bfloat16 else x.bfloat16()
        need_d = ctx.needs_input_grad[0]
        pf, pd = pack_conv_weight_pair(w.detach(), True, need_d, scale)               # w * scale, cast and both packings in one pass
        if res is not None:
            y = conv_igemm_residual(x, pf, shift, res, None, dil, k, relu)
        else:
            (y,) = conv_igemm([x], [pf], [shift], [dil], k, relu)
        ctx.save_for_backward(x, w, scale, y if relu else None)
        ctx.pd, ctx.dil, ctx.relu, ctx.k = pd, dil, relu, k
        ctx.link_in = link_in if (need_d and x.is_contiguous(memory_format=torch.channels_last)) else None
        ctx.link_out = link_out if relu else None
        if ctx.link_out is not None:
            ctx.link_out.scale, ctx.link_out.gb = 1.0, None
        ctx.has_res, ctx.res_link = res is not None, res_link
        if res_link is not None and res is None:
            res_link.armed, res_link.gm = bool(need_d), None     # the block's first node: it will add the shortcut's gradient itself
        return y

    @staticmethod
    def backward(ctx, g):
        from .ops import conv_igemm, conv_igemm_dgrad, conv_igemm_residual, conv_igemm_wgrad, conv_igemm_wgrad_launchable, relu_mask
        x, w, scale, y = ctx.saved_tensors
        cout, cin, k, d = w.shape[0], w.shape[1], ctx.k, ctx.dil
        cl = torch.channels_last
        g = g if g.dtype == torch.bfloat16 else g.bfloat16()
        masked = ctx.link_out.take(g) is not None if ctx.link_out is not None else False      # the consumer's data gradient came masked
        gm = (g if masked else relu_mask(g, y)) if ctx.relu else g
        gm = gm if gm.is_contiguous(memory_format=cl) else gm.contiguous(memory_format=cl)
        gres = None
        if ctx.has_res and ctx.needs_input_grad[8]:
            if ctx.res_link is not None and ctx.res_link.armed:
                ctx.res_link.gm = gm                                                    # the block's first node adds it in its store
            else:
                gres = gm
        shortcut = None
        if not ctx.has_res and ctx.res_link is not None:
            shortcut, ctx.res_link.gm = ctx.res_link.gm, None
        gx = None
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and _MERGED and conv_igemm_wgrad_launchable(cin, cout, k) and \
                x.is_contiguous(memory_format=cl):
            # the whole backward in one grid (the weight gradient's workgroups take the CUs the data gradient's tiles leave idle: a
            # 65 x 65 x 10 map is 166 pixel tiles), the scale on the weight gradient in its reduction, written into the reducer's slot
            from .backbone import _landed
            from .ops import conv_igemm_backward_residual
            from .reducer import grad_destination
            slot = grad_destination(w, w.shape)
            gx, gw = conv_igemm_backward_residual(gm, ctx.pd, x, d, k, x if ctx.link_in is not None else None, shortcut, scale, slot)
            if ctx.link_in is not None:
                ctx.link_in.leave(gx, True)
            return gx, _landed(gw, slot), None, None, None, None, None, None, gres, None
        if ctx.needs_input_grad[0]:
            if shortcut is not None:
                # the block's first convolution: data gradient + the gradient along the shortcut, and (x a block output with this
                # block as its one reader) the ReLU backward of the block in front — one store
                gx = conv_igemm_residual(gm, ctx.pd, None, shortcut, x if ctx.link_in is not None else None, d, k, False)
                if ctx.link_in is not None:
                    ctx.link_in.leave(gx, True)
            elif ctx.link_in is not None:
                # x = relu(...) of the node in front, read by this node only: its backward is a mask in this launch's store
                (gx,), _ = conv_igemm_dgrad([gm], [ctx.pd], [x], [d], k, 1.0, bias_grad=False)
                ctx.link_in.leave(gx, True)
            else:
                gx = conv_igemm([gm], [ctx.pd], None, [d], k, False)[0]
        gw = None
        if ctx.needs_input_grad[1]:
            if conv_igemm_wgrad_launchable(cin, cout, k):
                (gwf,) = conv_igemm_wgrad([x], [gm], [d], k)                    # float32, channels_last
            else:
                p = d * (k // 2)
                gwf = torch.ops.aten.convolution_backward(gm, x, w.to(torch.bfloat16), None, [1, 1], [p, p], [d, d], False, [0, 0], 1,
                                                          [False, True, False])[1].float()
            gw = gwf * scale.view(-1, 1, 1, 1)
        if shortcut is not None and gx is None:                                         # (cannot happen: the link is armed only by a node that computes one)
            raise RuntimeError("a shortcut gradient was left for a node that computes no data gradient")
        return gx, gw, None, None, None, None, None, None, gres, None


class _ResLink:
    """side channel inside one bottleneck, from its last node (the shortcut's add + ReLU in the store of the last convolution) to
    its first: the masked gradient of the block output, which reaches the block input along the identity shortcut and is added
    in the store of the first convolution's data gradient instead of by autograd's accumulation pass"""
    __slots__ = ("armed", "gm")

    def __init__(self):
        self.armed, self.gm = False, None


class _AddReLUFn(torch.autograd.Function):
    """relu(a + b) in one pass (ops.add_relu); both inputs receive the same masked gradient (one pass, ops.relu_mask)"""

    @staticmethod
    def forward(ctx, a, b):
        from .ops import add_relu
        y = add_relu(a, b)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        from .ops import relu_mask
        (y,) = ctx.saved_tensors
        gm = relu_mask(g if g.dtype == torch.bfloat16 else g.bfloat16(), y)
        return gm, gm


def _igemm_bn_route(x, conv, bn):
    """conv + frozen affine on the implicit-GEMM kernels?  bf16 CUDA activations, stride 1, 'same' padding, channel counts the
    forward AND the data-gradient launch take (conv_igemm: 64 | cin, 128 | cout, both ways round), a constant affine map"""
    from .ops import conv_igemm_supported
    k = conv.kernel_size[0]
    # (a 1x1 layer also with 64 channels on either side — res2's 256 -> 64 -> 256 at 129 x 129: half-empty tiles, but the layer is
    # bandwidth-bound and the library's GEMM for the shape runs 16 x 32 tiles, 203 us against ~40)
    takes = lambda ci, co: conv_igemm_supported(ci, co, k) or (k == 1 and co == 64 and ci % 64 == 0 and ci >= 64)      # noqa: E731
    return (x.is_cuda and conv.stride == (1, 1) and k in (1, 3) and conv.kernel_size == (k, k) and
            conv.padding[0] == conv.dilation[0] * (k // 2) and conv.padding[0] == conv.padding[1] and conv.groups == 1 and conv.bias is None and
            not bn.weight.requires_grad and not bn.bias.requires_grad and
            takes(conv.in_channels, conv.out_channels) and takes(conv.out_channels, conv.in_channels) and
            x.shape[0] * x.shape[2] * x.shape[3] >= 2048)


def _conv_bn(x, conv, bn, relu, link_in=None, link_out=None, res=None, res_link=None):
    """conv -> frozen-statistics BN (-> ReLU).  On the GPU, for stride-1 convolutions, the BN affine is folded into the
    weights (W * scale per output channel, bias = shift) and the whole thing is one im2col + GEMM with the bias (and ReLU)
    in the epilogue (backbone._ConvFn; 1x1 convolutions need no im2col at all).  gamma / beta still train: their
    gradients flow through the weight-sized products instead of activation-sized reductions."""
    if _IGEMM_BN and x.is_cuda and (x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)) \
            and _igemm_bn_route(x, conv, bn):
        scale, shift = bn.frozen_affine()
        return _FoldedIgemmFn.apply(x, conv.weight, scale, shift, conv.dilation[0], relu, link_in, link_out, res, res_link)
    if res is not None:
        raise RuntimeError("_conv_bn: a shortcut in the store needs the implicit-GEMM route")
    if x.is_cuda and conv.stride == (1, 1) and conv.kernel_size[0] in (1, 3) and conv.in_channels % 8 == 0 and \
            conv.padding[0] == conv.dilation[0] * (conv.kernel_size[0] // 2):
        scale = bn.weight * torch.rsqrt(bn.running_var + 1e-5)
        shift = bn.bias - bn.running_mean * scale
        return _ConvFn.apply(x, conv.weight * scale.view(-1, 1, 1, 1), shift, conv.dilation[0], relu, True, 0.0)
    y = bn(conv(x))
    return F.relu(y) if relu else y


class _Bottleneck(nn.Module):
    def __init__(self, cin, mid, stride, dilation, down, train_bn_affine=True):
        super().__init__()
        cout = mid * 4
        bn = lambda c: _FrozenBN(c, train_bn_affine)                      # noqa: E731
        self.c1, self.b1 = nn.Conv2d(cin, mid, 1, stride=stride, bias=False), bn(mid)
        self.c2, self.b2 = nn.Conv2d(mid, mid, 3, padding=dilation, dilation=dilation, bias=False), bn(mid)
        self.c3, self.b3 = nn.Conv2d(mid, cout, 1, bias=False), bn(cout)
        self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False), bn(cout)) if down else None

    def forward(self, x):
        # c1's and c2's outputs are ReLU outputs with one reader each (c2, c3): on the implicit-GEMM route their ReLU backward is a
        # mask in the store of the reader's data gradient (a _GradLink per pair, as in the VGG chain) instead of a pass of its own
        from .backbone import _GradLink, _FUSE_CHAIN
        l1, l2 = (_GradLink(), _GradLink()) if (_IGEMM_BN and _FUSE_CHAIN and torch.is_grad_enabled()) else (None, None)
        routed = lambda t, conv, bn: t.is_cuda and _igemm_bn_route(t, conv, bn)      # noqa: E731
        cl = torch.channels_last
        bf16 = x.is_cuda and (x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16))
        r1 = bf16 and _IGEMM_BN and routed(x, self.c1, self.b1)
        # the shortcut: relu(c3(..) + idn) in the store of c3's launch; with an identity shortcut its gradient joins c1's data gradient
        # in that launch's store (rl), and where x is itself such a block output (x._dsrg_plink, set below) the ReLU backward of the
        # block in front rides there too — no add / ReLU / mask / accumulate pass between the blocks, either way
        fuse = _IGEMM_BN and _FUSE_RES and bf16 and self.c3.stride == (1, 1)
        grad = torch.is_grad_enabled()
        rl = _ResLink() if (fuse and grad and _FUSE_CHAIN and self.down is None and r1 and x.dtype == torch.bfloat16) else None
        pin = getattr(x, "_dsrg_plink", None) if rl is not None else None
        y1 = _conv_bn(x, self.c1, self.b1, True, pin if r1 else None, l1 if l1 is not None and r1 else None, None, rl)
        use1 = l1 is not None and r1 and routed(y1, self.c2, self.b2)
        use2 = l2 is not None and routed(y1, self.c2, self.b2)
        y2 = _conv_bn(y1, self.c2, self.b2, True, l1 if use1 else None, l2 if use2 else None)
        use2 = use2 and routed(y2, self.c3, self.b3)
        idn = _conv_bn(x, self.down[0], self.down[1], False) if self.down is not None else x
        if fuse and routed(y2, self.c3, self.b3) and y2.dtype == torch.bfloat16 and idn.dtype == torch.bfloat16 and \
                idn.is_contiguous(memory_format=cl) and tuple(idn.shape) == (y2.shape[0], self.c3.out_channels, y2.shape[2], y2.shape[3]):
            pout = _GradLink() if (grad and _FUSE_CHAIN) else None
            y = _conv_bn(y2, self.c3, self.b3, True, l2 if use2 else None, pout, idn, rl)
            if pout is not None:
                y._dsrg_plink = pout                                      # the next block's first convolution may mask its data gradient with y
            return y
        if rl is not None:
            rl.armed = False                                              # (nobody will leave a gradient there)
        y = _conv_bn(y2, self.c3, self.b3, False, l2 if use2 else None, None)
        if _IGEMM_BN and y.is_cuda and y.dtype == torch.bfloat16 and idn.dtype == torch.bfloat16 and y.numel() % 8 == 0:
            return _AddReLUFn.apply(y, idn)                               # one pass each way instead of add + threshold
        return F.relu(y + idn)


class _AsppFn(torch.autograd.Function):
    """the DeepLab-v2 ASPP head, sum over four dilated 3x3 classifiers (2048 -> 21, dilation 6 / 12 / 18 / 24) of one feature map, as
    ONE 1x1 convolution on the implicit-GEMM kernels plus a shifted gather: out[p][o] = sum_j W_j[o] . f[p + off_j] over the 36
    (branch, tap) pairs = the gather of Y' = f x [all 36 x 21 tap kernels stacked] (ops.aspp_shift_sum).  No output channel padded
    from 21 to a 128-wide tile (the four-branch 3x3 launch multiplied 6x the real flops) and f is read once per pass instead of once per
    tap; backward: the gradient of Y' is the scatter of g (ops.aspp_shift_gather), the data and weight gradient of the 1x1 layer run in
    one grid (ops.conv_igemm_backward_residual).
    apply(dils, f, w_1..w_n, b_1..b_n): f bf16 channels_last; w, b the float32 master parameters -> (B, 21, H, W) float32."""

    @staticmethod
    def forward(ctx, dils, f, *t):
        from .ops import conv_igemm, pack_conv_weight_pair, aspp_shift_sum
        n = len(dils)
        ws, bs = t[:n], t[n:2 * n]
        O, cin = ws[0].shape[0], ws[0].shape[1]
        f = f if f.dtype == torch.bfloat16 else f.bfloat16()
        f = f if f.is_contiguous(memory_format=torch.channels_last) else f.contiguous(memory_format=torch.channels_last)
        J = 9 * n
        CT = (J * O + 127) // 128 * 128
        offsets = [((tap // 3 - 1) * d, (tap % 3 - 1) * d) for d in dils for tap in range(9)]
        # row j * O + o of the stacked kernel = w_branch[o, :, ky, kx], j = branch * 9 + ky * 3 + kx
        wcat = torch.cat([w.detach().float().permute(2, 3, 0, 1).reshape(9 * O, cin) for w in ws])
        wcat = F.pad(wcat, (0, 0, 0, CT - J * O)).view(CT, cin, 1, 1).contiguous(memory_format=torch.channels_last)
        need_d = ctx.needs_input_grad[1]
        pf, pd = pack_conv_weight_pair(wcat, True, need_d)
        (yp,) = conv_igemm([f], [pf], None, [1], 1, False)
        bias = bs[0].detach().float()
        for b in bs[1:]:
            bias = bias + b.detach().float()
        out = aspp_shift_sum(yp, offsets, O, bias.contiguous())
        ctx.save_for_backward(f)
        ctx.pd, ctx.offsets, ctx.O, ctx.CT, ctx.n, ctx.cin = pd, offsets, O, CT, n, cin
        return out

    @staticmethod
    def backward(ctx, g):
        from .ops import conv_igemm, conv_igemm_wgrad, conv_igemm_backward_residual, aspp_shift_gather
        (f,) = ctx.saved_tensors
        n, O, CT, cin = ctx.n, ctx.O, ctx.CT, ctx.cin
        gb = g.float().sum((0, 2, 3))                                                  # the same for every branch
        gp = aspp_shift_gather(g, ctx.offsets, CT)
        need_w = any(ctx.needs_input_grad[2:2 + n])
        gf = gw = None
        if ctx.needs_input_grad[1] and need_w:
            gf, gw = conv_igemm_backward_residual(gp, ctx.pd, f, 1, 1)
        elif ctx.needs_input_grad[1]:
            (gf,) = conv_igemm([gp], [ctx.pd], None, [1], 1, False)
        elif need_w:
            (gw,) = conv_igemm_wgrad([f], [gp], [1], 1)
        gws = [None] * n
        if gw is not None:
            g2 = gw.reshape(CT, cin)
            gws = [g2[b * 9 * O:(b + 1) * 9 * O].view(3, 3, O, cin).permute(2, 3, 0, 1) for b in range(n)]
        return (None, gf) + tuple(gws) + (gb,) * n


class ResNet101DeepLab(nn.Module):
    """DeepLab-v2 ResNet-101: output stride 8 (res4 dilation 2, res5 dilation 4), ASPP 6/12/18/24
    summed; 513x513 -> 65x65 (BASELINE.json configs[4])."""

    def __init__(self, num_classes=21, blocks=(3, 4, 23, 3), train_bn_affine=False):
        """train_bn_affine=False: the BatchNorm layers are constant affine maps (fixed statistics AND fixed gamma / beta, the usual
        DeepLab-v2 fine-tuning set-up), which lets the bottlenecks run on the implicit-GEMM kernels with the map folded in; True:
        gamma and beta train (their gradients flow through the folded weights), on the im2col + library-GEMM route"""
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), _FrozenBN(64, train_bn_affine), nn.ReLU(inplace=True),
                                  nn.MaxPool2d(3, 2, 1, ceil_mode=True))
        cfg = [(64, blocks[0], 1, 1), (128, blocks[1], 2, 1), (256, blocks[2], 1, 2), (512, blocks[3], 1, 4)]
        layers, cin = [], 64
        for mid, n, stride, dil in cfg:
            for i in range(n):
                layers.append(_Bottleneck(cin, mid, stride if i == 0 else 1, dil, down=(i == 0), train_bn_affine=train_bn_affine))
                cin = mid * 4
        self.layers = nn.Sequential(*layers)
        self.aspp = nn.ModuleList([nn.Conv2d(2048, num_classes, 3, padding=d, dilation=d) for d in (6, 12, 18, 24)])
        for m in self.aspp:
            nn.init.normal_(m.weight, std=0.01)
            nn.init.zeros_(m.bias)

    def forward(self, x):
        f = self.layers(self.stem(x))
        from .ops import conv_igemm_supported
        if _IGEMM_BN and f.is_cuda and (f.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)) \
                and len(self.aspp) <= 4 and 9 * len(self.aspp) * self.aspp[0].out_channels <= 2048 and conv_igemm_supported(f.shape[1], 128, 1) and \
                f.shape[1] % 256 == 0 and \
                f.shape[0] * f.shape[2] * f.shape[3] >= 2048 and all(m.kernel_size == (3, 3) and m.padding == m.dilation and m.stride == (1, 1)
                                                                      for m in self.aspp):
            return _AsppFn.apply(tuple(m.dilation[0] for m in self.aspp), f, *[m.weight for m in self.aspp], *[m.bias for m in self.aspp])
        out = self.aspp[0](f)
        for m in self.aspp[1:]:
            out = out + m(f)
        return out

    def caffe_param_groups(self):
        groups = {}
        for name, p in self.named_parameters():
            if not p.requires_grad:
                continue
            is_head, is_bias = name.startswith("aspp"), name.endswith("bias")
            key = ((10.0 if is_head else 1.0) * (2.0 if is_bias else 1.0), 0.0 if is_bias else 1.0)
            groups.setdefault(key, []).append(p)
        return [dict(params=ps, lr_mult=k[0], decay_mult=k[1]) for k, ps in groups.items()]


def count_flops_per_image_resnet101(size=513, blocks=(3, 4, 23, 3), num_classes=21):
    """forward multiply-accumulates x2 of ResNet101DeepLab's convolutions at size x size (output stride 8)"""
    def half(n): return (n - 1) // 2 + 1
    h = half(size)                                                   # 7x7 / 2 stem
    macs = 3 * 64 * 49 * h * h
    h = half(h)                                                      # 3x3 / 2 max pool (ceil mode)
    cin = 64
    for mid, n, stride in ((64, blocks[0], 1), (128, blocks[1], 2), (256, blocks[2], 1), (512, blocks[3], 1)):
        for i in range(n):
            ho = half(h) if (i == 0 and stride == 2) else h
            macs += cin * mid * ho * ho + mid * mid * 9 * ho * ho + mid * 4 * mid * ho * ho
            if i == 0:
                macs += cin * mid * 4 * ho * ho
            cin, h = mid * 4, ho
    macs += 4 * cin * num_classes * 9 * h * h
    return 2 * macs


class RetrainTrainer(object):
    """one train-f step: backbone -> (logits, labels shrunk by 8) -> softmax loss -> SGD with poly LR"""

    def __init__(self, device, world_size=1, backbone="vgg16", base_lr=1e-3, max_iter=20000, seed=0,
                 amp_dtype=torch.bfloat16, net=None, weights=None, snapshot=None, ddp=None):
        """weights: run.sh:9 `--weights models/model-s_iter_8000.caffemodel` — stage 2 starts from the stage-1 model
        (copied by layer name: .caffemodel / .npz / torch file); snapshot: a solverstate written by save()."""
        torch.manual_seed(seed)
        self.device, self.amp_dtype, self.max_iter, self.base_lr = device, amp_dtype, max_iter, base_lr
        net = net if net is not None else (VGG16ASPP() if backbone == "vgg16" else ResNet101DeepLab())
        if weights is not None:
            from .checkpoint import load_weights
            self.loaded_layers = load_weights(net, weights)
        net = net.to(device)
        if device.type == "cuda":
            net = net.to(memory_format=torch.channels_last)
        self.net = self.model = net
        self.reducer = None
        # the bias gradients' finishing passes as one launch behind backward: for the VGG backbone, whose nodes hand bias gradients on
        # untouched (the ResNet route's library-GEMM nodes are not checked for that)
        self.defer_bias = backbone == "vgg16" and torch.device(device).type == "cuda" and _os.environ.get("DSRG_DEFER_REDUCTIONS", "1") != "0"
        if (world_size > 1) if ddp is None else ddp:
            from .reducer import BucketedAllReduce           # as DSRGTrainer: gradients land in their all-reduce buckets
            self.reducer = BucketedAllReduce(list(net.parameters()), bucket_cap_mb=32)
        self.opt = CaffeSGD(net.caffe_param_groups(), base_lr=base_lr, gamma=1.0, stepsize=1 << 30)
        if snapshot is not None:
            self.load(snapshot)

    def save(self, prefix="models/model-f"):
        """solver-f.prototxt:15-16 `snapshot_prefix`"""
        from .checkpoint import save_snapshot
        return save_snapshot(self, prefix)

    def load(self, state_path):
        from .checkpoint import load_snapshot
        return load_snapshot(self, state_path)          # the poly rate is recomputed from opt.iter at every step

    def step(self, images, label):
        if self.reducer is not None:
            self.reducer.prepare()
        else:
            self.opt.zero_grad()
        self.opt.base_lr = poly_lr(self.base_lr, self.opt.iter, self.max_iter)
        x = images.contiguous(memory_format=torch.channels_last) if self.device.type == "cuda" else images
        with torch.autocast(self.device.type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            logits = self.model(x)
        loss = seg_softmax_loss(logits, interp_shrink(label, 8))
        from .ops import deferred_reductions
        with deferred_reductions(self.defer_bias):             # (see DSRGTrainer.step)
            loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
        self.opt.step()
        return loss.detach()
