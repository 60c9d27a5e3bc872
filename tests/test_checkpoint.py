"""CPU: weights in / snapshots out (dsrg_amd/checkpoint.py) — what `train.py --weights X.caffemodel`, `--snapshot` and
the solver's `snapshot_prefix` do in the reference (tools/train.py:53-63, run.sh:5,9, solver-s.prototxt:16-17)."""
import os

import numpy as np
import pytest
import torch

from dsrg_amd import checkpoint as CK
from dsrg_amd.backbone import VGG16ASPP
from dsrg_amd.trainer import CaffeSGD, DSRGTrainer
from test_data_parallel import TinyNet, torch_loss, make_data


def test_caffemodel_codec_roundtrip_and_legacy_forms(tmp_path):
    rng = np.random.default_rng(0)
    layers = {"conv1_1": [rng.standard_normal((4, 3, 3, 3)).astype(np.float32), rng.standard_normal(4).astype(np.float32)],
              "fc8-SEC_2": [rng.standard_normal((21, 8, 1, 1)).astype(np.float32), np.zeros(21, np.float32)],
              "scalarish": [np.float32([1.5])]}
    p = str(tmp_path / "m.caffemodel")
    CK.write_caffemodel(p, layers)
    got = CK.read_caffemodel(p)
    assert list(got) == list(layers)
    for k in layers:
        assert len(got[k]) == len(layers[k])
        for a, b in zip(got[k], layers[k]):
            assert a.shape == b.shape and np.array_equal(a, b)
    # the legacy encodings a 2015-era vgg16_20M_mc.caffemodel may use: V1 `layers` (field 2, name = 4, blobs = 6),
    # num/channels/height/width instead of a BlobShape, unpacked floats
    w = np.arange(6, dtype=np.float32).reshape(1, 2, 1, 3)
    blob = b"".join(CK._enc_varint((f << 3) | 0) + CK._enc_varint(d) for f, d in zip((1, 2, 3, 4), w.shape))
    blob += b"".join(CK._enc_varint((5 << 3) | 5) + np.float32(v).tobytes() for v in w.ravel())
    lay = CK._enc_ld(4, b"old_conv") + CK._enc_ld(6, blob)
    open(p, "wb").write(CK._enc_ld(1, b"net") + CK._enc_ld(2, lay))
    got = CK.read_caffemodel(p)
    assert list(got) == ["old_conv"] and got["old_conv"][0].shape == (1, 2, 1, 3) and np.array_equal(got["old_conv"][0], w)


def test_vgg_weights_by_caffe_layer_name(tmp_path):
    torch.manual_seed(0)
    a = VGG16ASPP()
    lm = CK.caffe_layer_map(a)
    assert len(lm) == 13 + 12 and lm["conv5_3"] is a.features[28] and lm["fc8-SEC_3"] is a.branches[2][6]
    assert lm["fc6_2"].dilation == (12, 12) and lm["fc7_4"].kernel_size == (1, 1)
    for ext in (".caffemodel", ".npz", ".pt"):
        p = str(tmp_path / ("w" + ext))
        CK.save_weights(a, p)
        torch.manual_seed(1)
        b = VGG16ASPP()
        copied = CK.load_weights(b, p)
        assert len(copied) >= 25
        for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
            assert ka == kb and torch.equal(va, vb), (ext, ka)
    # the ImageNet init has no fc8-SEC heads: they keep their N(0, 0.01) initialisation, everything else is copied
    layers = {k: v for k, v in CK.net_to_layers(a).items() if not k.startswith("fc8")}
    p = str(tmp_path / "init.caffemodel")
    CK.write_caffemodel(p, layers)
    torch.manual_seed(2)
    c = VGG16ASPP()
    head0 = c.branches[0][6].weight.clone()
    copied = CK.load_weights(c, p)
    assert "fc8-SEC_1" not in copied and torch.equal(c.branches[0][6].weight, head0)
    assert torch.equal(c.features[0].weight, a.features[0].weight) and torch.equal(c.branches[3][3].bias, a.branches[3][3].bias)
    # a shape mismatch is an error, as in Caffe
    layers["conv1_1"][0] = layers["conv1_1"][0][:, :2]
    CK.write_caffemodel(p, layers)
    try:
        CK.load_weights(VGG16ASPP(), p)
        assert False, "shape mismatch must raise"
    except ValueError:
        pass


def test_snapshot_resume_continues_the_same_trajectory(tmp_path):
    """5 steps straight == 3 steps, snapshot, new process state, restore, 2 steps (momentum history, lr schedule and
    iteration come back); the .caffemodel next to it initialises a stage-2 trainer by layer name"""
    dev = torch.device("cpu")
    images, labels, cues = make_data(4)

    def make():
        torch.manual_seed(3)                     # TinyNet() draws its weights before the trainer seeds
        t = DSRGTrainer(dev, amp_dtype=None, channels_last=False, loss_fn=torch_loss, net=TinyNet(), seed=3)
        t.opt.stepsize = 2                       # the rate steps inside the run
        return t
    a = make()
    for _ in range(5):
        a.step(images, labels, cues)
    b = make()
    for _ in range(3):
        b.step(images, labels, cues)
    model_path, state_path = b.save(str(tmp_path / "models" / "model-s"))
    assert model_path.endswith("model-s_iter_3.caffemodel") and os.path.exists(model_path) and os.path.exists(state_path)
    c = DSRGTrainer(dev, amp_dtype=None, channels_last=False, loss_fn=torch_loss, net=TinyNet(), seed=99, snapshot=state_path)
    c.opt.stepsize = 2
    assert c.opt.iter == 3
    for _ in range(2):
        c.step(images, labels, cues)
    for (k, va), vc in zip(a.net.state_dict().items(), c.net.state_dict().values()):
        assert torch.allclose(va, vc, rtol=0, atol=1e-7), k
    d = DSRGTrainer(dev, amp_dtype=None, channels_last=False, loss_fn=torch_loss, net=TinyNet(), seed=5, weights=model_path)
    assert len(d.loaded_layers) == 7
    for vb, vd in zip(b.net.state_dict().values(), d.net.state_dict().values()):
        assert torch.equal(vb, vd)
    assert d.opt.iter == 0                       # --weights does not restore the solver


def test_fused_pool_keeps_the_sequential_slots_and_the_cpu_arithmetic():
    """pool1-3 run inside the convolution node in front of them on the GPU (GemmConv2d(fuse_pool=...)); the Sequential keeps a
    placeholder in the pool's slot, so state_dict keys and the Caffe layer map are those of the plain conv / ReLU / pool stack,
    and on the CPU the module is conv -> ReLU -> 3x3 / stride 2 / pad 1 ceil-mode max pool"""
    import torch.nn.functional as F
    from dsrg_amd.backbone import VGG16ASPP, GemmConv2d, FusedPool, MaxPool3x3
    from dsrg_amd.checkpoint import caffe_layer_map
    torch.manual_seed(0)
    net = VGG16ASPP()
    kinds = [type(m).__name__ for m in net.features]
    assert len(kinds) == 32 and kinds[4] == kinds[9] == kinds[16] == "FusedPool" and kinds.count("MaxPool3x3") == 2
    conv_slots = [i for i, m in enumerate(net.features) if isinstance(m, torch.nn.Conv2d)]
    assert conv_slots == [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]            # train-s.prototxt's conv1_1 ... conv5_3
    assert [net.features[i].fuse_pool for i in (2, 7, 14)] == [(2, True)] * 3
    lm = caffe_layer_map(net)
    assert lm["conv1_2"] is net.features[2] and lm["conv3_3"] is net.features[14] and lm["conv5_3"] is net.features[28]
    a = GemmConv2d(8, 16, 3, padding=1, fuse_relu=True, fuse_pool=(2, True))
    x = torch.randn(2, 8, 13, 10)
    want = F.max_pool2d(F.relu(F.conv2d(x, a.weight, a.bias, padding=1)), 3, 2, 1, ceil_mode=True)
    assert torch.equal(a(x), want)
    with pytest.raises(ValueError):
        GemmConv2d(8, 16, 3, padding=1, fuse_pool=(2, True))                          # no ReLU in front of the pool
