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

from .backbone import VGG16ASPP, GemmConv2d, _ConvFn
from .trainer import CaffeSGD


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
    """BatchNorm with fixed statistics (DeepLab-v2 trains ResNet-101 with use_global_stats)"""

    def __init__(self, c):
        super().__init__()
        self.weight, self.bias = nn.Parameter(torch.ones(c)), nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))

    def forward(self, x):
        scale = self.weight * torch.rsqrt(self.running_var + 1e-5)
        shift = self.bias - self.running_mean * scale
        return x * scale.view(1, -1, 1, 1).to(x.dtype) + shift.view(1, -1, 1, 1).to(x.dtype)


def _conv_bn(x, conv, bn, relu):
    """conv -> frozen-statistics BN (-> ReLU).  On the GPU, for stride-1 convolutions, the BN affine is folded into the
    weights (W * scale per output channel, bias = shift) and the whole thing is one im2col + GEMM with the bias (and ReLU)
    in the epilogue (backbone._ConvFn; 1x1 convolutions need no im2col at all).  gamma / beta still train: their
    gradients flow through the weight-sized products instead of activation-sized reductions."""
    if x.is_cuda and conv.stride == (1, 1) and conv.kernel_size[0] in (1, 3) and conv.in_channels % 8 == 0 and \
            conv.padding[0] == conv.dilation[0] * (conv.kernel_size[0] // 2):
        scale = bn.weight * torch.rsqrt(bn.running_var + 1e-5)
        shift = bn.bias - bn.running_mean * scale
        return _ConvFn.apply(x, conv.weight * scale.view(-1, 1, 1, 1), shift, conv.dilation[0], relu, True, 0.0)
    y = bn(conv(x))
    return F.relu(y) if relu else y


class _Bottleneck(nn.Module):
    def __init__(self, cin, mid, stride, dilation, down):
        super().__init__()
        cout = mid * 4
        self.c1, self.b1 = nn.Conv2d(cin, mid, 1, stride=stride, bias=False), _FrozenBN(mid)
        self.c2, self.b2 = nn.Conv2d(mid, mid, 3, padding=dilation, dilation=dilation, bias=False), _FrozenBN(mid)
        self.c3, self.b3 = nn.Conv2d(mid, cout, 1, bias=False), _FrozenBN(cout)
        self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False), _FrozenBN(cout)) if down else None

    def forward(self, x):
        y = _conv_bn(x, self.c1, self.b1, True)
        y = _conv_bn(y, self.c2, self.b2, True)
        y = _conv_bn(y, self.c3, self.b3, False)
        idn = _conv_bn(x, self.down[0], self.down[1], False) if self.down is not None else x
        return F.relu(y + idn)


class ResNet101DeepLab(nn.Module):
    """DeepLab-v2 ResNet-101: output stride 8 (res4 dilation 2, res5 dilation 4), ASPP 6/12/18/24
    summed; 513x513 -> 65x65 (BASELINE.json configs[4])."""

    def __init__(self, num_classes=21, blocks=(3, 4, 23, 3)):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), _FrozenBN(64), nn.ReLU(inplace=True),
                                  nn.MaxPool2d(3, 2, 1, ceil_mode=True))
        cfg = [(64, blocks[0], 1, 1), (128, blocks[1], 2, 1), (256, blocks[2], 1, 2), (512, blocks[3], 1, 4)]
        layers, cin = [], 64
        for mid, n, stride, dil in cfg:
            for i in range(n):
                layers.append(_Bottleneck(cin, mid, stride if i == 0 else 1, dil, down=(i == 0)))
                cin = mid * 4
        self.layers = nn.Sequential(*layers)
        self.aspp = nn.ModuleList([nn.Conv2d(2048, num_classes, 3, padding=d, dilation=d) for d in (6, 12, 18, 24)])
        for m in self.aspp:
            nn.init.normal_(m.weight, std=0.01)
            nn.init.zeros_(m.bias)

    def forward(self, x):
        f = self.layers(self.stem(x))
        out = self.aspp[0](f)
        for m in self.aspp[1:]:
            out = out + m(f)
        return out

    def caffe_param_groups(self):
        groups = {}
        for name, p in self.named_parameters():
            is_head, is_bias = name.startswith("aspp"), name.endswith("bias")
            key = ((10.0 if is_head else 1.0) * (2.0 if is_bias else 1.0), 0.0 if is_bias else 1.0)
            groups.setdefault(key, []).append(p)
        return [dict(params=ps, lr_mult=k[0], decay_mult=k[1]) for k, ps in groups.items()]


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
        if (world_size > 1) if ddp is None else ddp:
            from torch.nn.parallel import DistributedDataParallel as DDP
            self.model = DDP(net, device_ids=[device.index] if device.type == "cuda" else None, bucket_cap_mb=32,
                             gradient_as_bucket_view=True)
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
        self.opt.zero_grad()
        self.opt.base_lr = poly_lr(self.base_lr, self.opt.iter, self.max_iter)
        x = images.contiguous(memory_format=torch.channels_last) if self.device.type == "cuda" else images
        with torch.autocast(self.device.type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None):
            logits = self.model(x)
        loss = seg_softmax_loss(logits, interp_shrink(label, 8))
        loss.backward()
        self.opt.step()
        return loss.detach()
