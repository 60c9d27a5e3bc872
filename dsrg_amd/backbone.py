"""VGG16 DeepLab-v2 ASPP backbone of training/experiment/seed_mc/train-s.prototxt:41-744.

Plumbing around the hot path: the convolutions run through PyTorch-ROCm (MIOpen /
hipBLASLt, MFMA bf16 under autocast) — the only GEMM-shaped part of a DSRG step.
Layer shapes follow the prototxt exactly: 3x3/2 ceil-mode max pools with pad 1
(321 -> 161 -> 81 -> 41), pool4/pool5 stride 1, conv5_x dilation 2, AVE pool5a,
four branches fc6_k (3x3, dilation 6/12/18/24) -> fc7_k (1x1) -> fc8-SEC_k (1x1, 21
outputs, N(0, 0.01) init), summed (Eltwise SUM, train-s.prototxt:737-744).
"""
import torch
import torch.nn as nn


def _conv_relu(cin, cout, dilation=1):
    return [nn.Conv2d(cin, cout, 3, padding=dilation, dilation=dilation), nn.ReLU(inplace=True)]


class VGG16ASPP(nn.Module):
    def __init__(self, num_classes=21, dropout=0.5):
        super().__init__()
        L = []
        L += _conv_relu(3, 64) + _conv_relu(64, 64) + [nn.MaxPool2d(3, 2, 1, ceil_mode=True)]
        L += _conv_relu(64, 128) + _conv_relu(128, 128) + [nn.MaxPool2d(3, 2, 1, ceil_mode=True)]
        L += _conv_relu(128, 256) + _conv_relu(256, 256) + _conv_relu(256, 256) + [nn.MaxPool2d(3, 2, 1, ceil_mode=True)]
        L += _conv_relu(256, 512) + _conv_relu(512, 512) + _conv_relu(512, 512) + [nn.MaxPool2d(3, 1, 1)]
        L += _conv_relu(512, 512, 2) + _conv_relu(512, 512, 2) + _conv_relu(512, 512, 2) + [nn.MaxPool2d(3, 1, 1)]
        L += [nn.AvgPool2d(3, 1, 1)]                                   # pool5a AVE (count_include_pad, as Caffe)
        self.features = nn.Sequential(*L)
        self.branches = nn.ModuleList()
        for d in (6, 12, 18, 24):
            fc8 = nn.Conv2d(1024, num_classes, 1)
            nn.init.normal_(fc8.weight, std=0.01)
            nn.init.zeros_(fc8.bias)
            self.branches.append(nn.Sequential(
                nn.Conv2d(512, 1024, 3, padding=d, dilation=d), nn.ReLU(inplace=True), nn.Dropout(dropout),
                nn.Conv2d(1024, 1024, 1), nn.ReLU(inplace=True), nn.Dropout(dropout), fc8))

    def forward(self, x):
        f = self.features(x)
        out = self.branches[0](f)
        for br in self.branches[1:]:
            out = out + br(f)
        return out

    def caffe_param_groups(self):
        """lr_mult / decay_mult of the prototxt: weights (1,1), biases (2,0); fc8-SEC (10,1)/(20,0)."""
        groups = {}
        for name, p in self.named_parameters():
            is_fc8 = name.startswith("branches") and name.split(".")[2] == "6"
            is_bias = name.endswith("bias")
            key = ((10.0 if is_fc8 else 1.0) * (2.0 if is_bias else 1.0), 0.0 if is_bias else 1.0)
            groups.setdefault(key, []).append(p)
        return [dict(params=ps, lr_mult=k[0], decay_mult=k[1]) for k, ps in groups.items()]


def count_flops_per_image(size=321):
    """forward multiply-accumulates x2 of the conv stack at size x size (160.2 GFLOP)."""
    def pool(n): return -(-(n + 2 - 3) // 2) + 1
    h1 = size; h2 = pool(h1); h3 = pool(h2); h4 = pool(h3)
    macs = 0
    for (cin, cout, h, n) in [(3, 64, h1, 1), (64, 64, h1, 1), (64, 128, h2, 1), (128, 128, h2, 1),
                              (128, 256, h3, 1), (256, 256, h3, 2), (256, 512, h4, 1), (512, 512, h4, 5)]:
        macs += n * cin * cout * 9 * h * h
    macs += 4 * (512 * 1024 * 9 + 1024 * 1024 + 1024 * 21) * h4 * h4
    return 2 * macs
