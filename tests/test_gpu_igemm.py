"""GPU (-m gpu): the implicit-GEMM convolution kernels of the backbone (csrc/conv_igemm.hip) against torch in fp32 on the
same bf16-valued operands — backbone plumbing (train-s.prototxt:161-736 are Caffe Convolution layers; no oracle)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
CL = torch.channels_last


@pytest.fixture(scope="module")
def ops():
    from dsrg_amd import ops, _lib
    _lib.require_gpu()
    return ops


def _case(B, H, W, cin, cout, k, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(B, cin, H, W, device="cuda", generator=g).bfloat16().contiguous(memory_format=CL)
    w = (torch.randn(cout, cin, k, k, device="cuda", generator=g) * (2.0 / (cin * k * k)) ** 0.5).bfloat16()
    b = torch.randn(cout, device="cuda", generator=g)
    return x, w, b


def _close(got, want, what):
    err = (got.float() - want).abs().max()
    assert err <= 0.01 * want.abs().max() + 1e-3, (what, float(err), float(want.abs().max()))


@pytest.mark.parametrize("B,H,W,cin,cout,k,dil,relu,bias", [
    (1, 5, 7, 64, 256, 3, 1, True, True),          # one partial tile, every border case
    (1, 5, 7, 64, 256, 3, 9, False, False),        # dilation beyond the map: only the centre tap is inside
    (2, 19, 23, 128, 256, 3, 2, True, True),       # several tiles, a ragged last one, two channel chunks
    (1, 41, 41, 512, 512, 3, 2, True, True),       # conv5_x
    (2, 41, 41, 512, 1024, 3, 12, True, True),     # fc6_2
    (1, 41, 41, 1024, 512, 3, 6, False, False),    # its data gradient's shape
    (2, 41, 41, 1024, 1024, 1, 1, True, True),     # fc7
    (1, 81, 81, 256, 256, 3, 1, True, True),       # conv3_2
    (3, 1, 1, 64, 256, 3, 1, True, True),
    (2, 27, 31, 256, 128, 3, 1, False, False),     # conv3_1's data gradient: half an n-tile (waves 4-7 only move data)
    (1, 20, 20, 128, 384, 3, 2, True, True),       # one and a half n-tiles
])
def test_igemm_conv_matches_torch(ops, B, H, W, cin, cout, k, dil, relu, bias):
    x, w, b = _case(B, H, W, cin, cout, k, 5)
    want = F.conv2d(x.float(), w.float(), b if bias else None, padding=dil * (k // 2), dilation=dil)
    want = torch.relu(want) if relu else want
    packed = ops.pack_conv_weight(w)
    for variant in (1, 2, 5):
        ops.set_igemm_variant(variant)
        (got,) = ops.conv_igemm([x], [packed], [b if bias else None], [dil], k, relu)
        assert got.shape == want.shape and got.dtype == torch.bfloat16 and got.is_contiguous(memory_format=CL)
        _close(got, want, "variant %d" % variant)
        (again,) = ops.conv_igemm([x], [packed], [b if bias else None], [dil], k, relu)
        assert torch.equal(got, again)                                               # deterministic
    ops.set_igemm_variant(1)


def test_igemm_packed_from_fp32_master_and_data_gradient(ops):
    """pack_conv_weight casts the fp32 master weights in its one copy; with for_dgrad the same kernel is the data gradient"""
    B, H, W, cin, cout, dil = 2, 13, 17, 256, 512, 2
    x, w, _ = _case(B, H, W, cin, cout, 3, 7)
    w32 = w.float() + 1e-4 * torch.randn_like(w.float())                             # not bf16-representable
    assert torch.equal(ops.pack_conv_weight(w32), ops.pack_conv_weight(w32.bfloat16()))
    g = torch.randn(B, cout, H, W, device="cuda").bfloat16().contiguous(memory_format=CL)
    xr = x.float().requires_grad_(True)
    F.conv2d(xr, w.float(), None, padding=dil, dilation=dil).backward(g.float())
    (gx,) = ops.conv_igemm([g], [ops.pack_conv_weight(w, for_dgrad=True)], [None], [dil], 3, False)
    assert gx.shape == x.shape
    _close(gx, xr.grad, "dgrad")


def test_igemm_four_branches_share_a_launch(ops):
    """the ASPP form: four problems (own input, kernel, bias, dilation) in one launch == four launches"""
    B, H, W, cin, cout = 1, 41, 41, 512, 1024
    xs, ws, bs, dils = [], [], [], [6, 12, 18, 24]
    for i in range(4):
        x, w, b = _case(B, H, W, cin, cout, 3, 20 + i)
        xs.append(x); ws.append(ops.pack_conv_weight(w)); bs.append(b)
    together = ops.conv_igemm(xs, ws, bs, dils, 3, True)
    for i in range(4):
        (alone,) = ops.conv_igemm([xs[i]], [ws[i]], [bs[i]], [dils[i]], 3, True)
        assert torch.equal(together[i], alone)
    # same input for every branch (fc6_k all read pool5a)
    shared = ops.conv_igemm([xs[0]] * 4, ws, bs, dils, 3, True)
    want = torch.relu(F.conv2d(xs[0].float(), _unpack(ws[2]), bs[2], padding=18, dilation=18))
    _close(shared[2], want, "shared input")


@pytest.mark.parametrize("B,H,W", [(1, 41, 41), (3, 41, 41), (2, 65, 65), (2, 30, 50)])
def test_igemm_border_taps_are_skipped_bit_identically(ops, B, H, W):
    """A dilated tap that reaches no pixel of a 256-pixel tile (forward / data gradient) or no row of a 64-pixel step (weight
    gradient) multiplies padding zeros only; those K-steps are not loaded and not multiplied (conv_igemm.hip).  Same bits as
    with every step multiplied (variant 6: the launch of rounds 4), four branches, dilations 6 .. 24 — and right against torch."""
    cin, cout = 512, 1024
    dils = [6, 12, 18, 24]
    xs, ws, ws32, bs = [], [], [], []
    for i in range(4):
        x, w, b = _case(B, H, W, cin, cout, 3, 50 + i)
        xs.append(x); ws.append(ops.pack_conv_weight(w)); ws32.append(w); bs.append(b)
    gs = [torch.randn(B, cout, H, W, device="cuda").bfloat16().contiguous(memory_format=CL) for _ in range(4)]
    try:
        ops.set_igemm_variant(6)
        full = ops.conv_igemm(xs, ws, bs, dils, 3, True, stream_k=False)      # (stream-K cuts the K range: another summation order)
        full_w = ops.conv_igemm_wgrad([xs[0]] * 4, gs, dils, 3)
        ops.set_igemm_variant(7)                                 # the weight gradient skips dead 64-pixel steps of the flat pixel order
        flat_w = ops.conv_igemm_wgrad([xs[0]] * 4, gs, dils, 3)
        ops.set_igemm_variant(3)                                 # ... or sums over the pixels each tap reaches only (the default)
        skip = ops.conv_igemm(xs, ws, bs, dils, 3, True, stream_k=False)
        skip_w = ops.conv_igemm_wgrad([xs[0]] * 4, gs, dils, 3)
    finally:
        ops.set_igemm_variant(-1)
    for i in range(4):
        assert torch.equal(full[i], skip[i]), dils[i]
        assert torch.equal(full_w[i], flat_w[i]), dils[i]
        # the compact reduction adds the same non-zero terms in the same order but cuts them into other partial sums (every
        # split takes an equal share of the LIVE pixels): equal up to fp32 reassociation
        assert (full_w[i] - skip_w[i]).abs().max() <= 2e-5 * full_w[i].abs().max(), dils[i]
    i = 3
    want = torch.relu(F.conv2d(xs[i].float(), ws32[i].float(), bs[i], padding=dils[i], dilation=dils[i]))
    _close(skip[i], want, "dilation 24 forward")
    wr = torch.zeros(cout, cin, 3, 3, device="cuda", requires_grad=True)
    F.conv2d(xs[0].float(), wr, None, padding=dils[i], dilation=dils[i]).backward(gs[i].float())
    assert (skip_w[i] - wr.grad).abs().max() <= 2e-3 * wr.grad.abs().max() + 1e-4


def test_igemm_skipping_on_random_shapes(ops):
    """the tap / row skipping and the XCD-interleaved tile maps on shapes nobody tuned for: random maps (1 .. 70 a side), batch
    sizes, dilations up to beyond the map (every tap but the centre dead), one to four groups — fewer pixel tiles than XCDs,
    tiles that straddle images, pixel chunks of a single step: forward, data-gradient form and weight gradient must equal the
    every-step launch bit for bit, and the forward must be right against torch"""
    g = torch.Generator().manual_seed(99)
    rnd = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))          # noqa: E731
    for case in range(24):
        B, H, W = rnd(1, 5), rnd(1, 70), rnd(1, 70)
        n = rnd(1, 4)
        dils = [rnd(3, 40) if case % 3 else rnd(1, 12) for _ in range(n)]
        cin, cout = (256, 256) if case % 2 else (256, 512)
        xs = [torch.randn(B, cin, H, W, device="cuda").bfloat16().contiguous(memory_format=CL) for _ in range(n)]
        w32 = [(torch.randn(cout, cin, 3, 3, device="cuda") * 0.03).bfloat16() for _ in range(n)]
        ws = [ops.pack_conv_weight(w) for w in w32]
        gs = [torch.randn(B, cout, H, W, device="cuda").bfloat16().contiguous(memory_format=CL) for _ in range(n)]
        try:
            ops.set_igemm_variant(6)
            full = ops.conv_igemm(xs, ws, [None] * n, dils, 3, False, stream_k=False)
            full_w = ops.conv_igemm_wgrad(xs, gs, dils, 3)
            ops.set_igemm_variant(7)
            flat_w = ops.conv_igemm_wgrad(xs, gs, dils, 3)
            ops.set_igemm_variant(3)
            skip = ops.conv_igemm(xs, ws, [None] * n, dils, 3, False, stream_k=False)
            skip_w = ops.conv_igemm_wgrad(xs, gs, dils, 3)
        finally:
            ops.set_igemm_variant(-1)
        for i in range(n):
            assert torch.equal(full[i], skip[i]), (case, B, H, W, dils, i)
            assert torch.equal(full_w[i], flat_w[i]), (case, B, H, W, dils, i)
            assert (full_w[i] - skip_w[i]).abs().max() <= 2e-5 * full_w[i].abs().max() + 1e-6, (case, B, H, W, dils, i)
        want = F.conv2d(xs[0].float(), w32[0].float(), None, padding=dils[0], dilation=dils[0])
        _close(skip[0], want, "case %d: %dx%dx%d dilation %d" % (case, B, H, W, dils[0]))


@pytest.mark.parametrize("B,H,W,cin,cout,k,dil", [(2, 41, 41, 128, 256, 3, 1), (1, 33, 29, 64, 128, 3, 6), (2, 17, 19, 256, 128, 1, 1)])
def test_igemm_split_mode_is_float32_grade(ops, B, H, W, cin, cout, k, dil):
    """dsrg_conv_igemm_split_f32 (the layer-level prototype of a float32 convolution on the bf16 MFMA: operands as three bf16
    planes, six bf16 products per multiply-add on the fp32 accumulators): against a float64 convolution of the same float32
    operands the error is that of a float32 convolution (1e-6 of the range), three orders below the bf16 launch's"""
    g = torch.Generator(device="cuda").manual_seed(7 + cin)
    x = torch.relu(torch.randn(B, cin, H, W, device="cuda", generator=g)).contiguous(memory_format=CL)
    w = torch.randn(cout, cin, k, k, device="cuda", generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, device="cuda", generator=g) * 0.1
    y64 = F.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), padding=dil * (k // 2), dilation=dil)
    for relu in (False, True):
        y = ops.conv_igemm_split(x, w, b, dil, relu)
        want = torch.relu(y64) if relu else y64
        assert y.dtype == torch.float32 and y.is_contiguous(memory_format=CL)
        err = float((y.double().cpu() - want).abs().max()) / float(y64.abs().max())
        y32 = F.conv2d(x.cpu(), w.cpu(), b.cpu(), padding=dil * (k // 2), dilation=dil)
        err32 = float((y32.double() - y64).abs().max()) / float(y64.abs().max())
        assert err < 4e-6 and err < 4 * err32 + 1e-6, (err, err32)
    yb = ops.conv_igemm([x.bfloat16().contiguous(memory_format=CL)], [ops.pack_conv_weight(w.bfloat16())], [b], [dil], k, False)[0]
    assert float((yb.double().cpu() - y64).abs().max()) / float(y64.abs().max()) > 100 * err


@pytest.mark.parametrize("B,H,W", [(16, 41, 41), (1, 41, 41), (3, 65, 65), (2, 30, 50), (5, 7, 100), (2, 23, 9)])
def test_igemm_class_ordered_tiles_are_bit_identical(ops, B, H, W):
    """The dilated launches order their pixels class by class (the <= 3 x 3 rectangles of the map inside each of which every
    pixel has the same live taps, most taps first) and cut THAT order into tiles of 256, so that a tile multiplies padding
    only where it straddles two classes (conv_igemm.hip, IgemmArgs::cls_tiles; variant 9 forces it, 8 = round 5's row-aligned
    tiles, 6 = every step of flat tiles).  Every output element still sums its live taps in the same order: forward (bias,
    ReLU, Dropout: the mask is a function of the PIXEL, not of the tile row) and the masked data gradient are the same bits;
    the bias gradient sums other partial rows (fp32 reassociation)."""
    cin, cout = 256, 512
    dils = [6, 12, 18, 24] if (H, W) != (23, 9) else [3, 5, 30, 4]
    xs, ws, bs = [], [], []
    for i in range(4):
        x, w, b = _case(B, H, W, cin, cout, 3, 80 + i)
        xs.append(x); ws.append(ops.pack_conv_weight(w)); bs.append(b)
    gs = [torch.randn(B, cout, H, W, device="cuda").bfloat16().contiguous(memory_format=CL) for _ in range(4)]
    wd = [ops.pack_conv_weight(torch.randn(cout, cin, 3, 3, device="cuda") * 0.02, for_dgrad=True) for _ in range(4)]
    ys = [torch.relu(torch.randn(B, cin, H, W, device="cuda")).bfloat16().contiguous(memory_format=CL) for _ in range(4)]
    out = {}
    try:
        for v in (6, 8, 9, 3):
            ops.set_igemm_variant(v)
            f = ops.conv_igemm(xs, ws, bs, dils, 3, True, 0.5, 4321, stream_k=False)
            d, gb = ops.conv_igemm_dgrad(gs, wd, ys, dils, 3, 2.0)
            out[v] = (f, d, gb)
    finally:
        ops.set_igemm_variant(-1)
    for v in (8, 9, 3):
        for i in range(4):
            assert torch.equal(out[6][0][i], out[v][0][i]), (v, i)
            assert torch.equal(out[6][1][i], out[v][1][i]), (v, i)
            assert float((out[6][2][i] - out[v][2][i]).abs().max()) <= 1e-4 * float(out[6][2][i].abs().max()) + 1e-5, (v, i)
    want = torch.relu(F.conv2d(xs[1].float(), _unpack(ws[1]), bs[1], padding=dils[1], dilation=dils[1]))
    kept = out[9][0][1] != 0
    assert ((out[9][0][1].float() - 2.0 * want)[kept].abs() <= 0.02 * want[kept].abs() + 2e-2).all()


@pytest.mark.parametrize("B,H,W,cin,cout,dil,absorb", [
    (16, 41, 41, 512, 512, 1, True),      # conv4_2 / conv4_3: 212 data-gradient tiles + 252 weight-gradient workgroups in one grid
    (2, 41, 41, 512, 512, 2, True),       # conv5_x
    (3, 41, 41, 256, 512, 1, False),      # conv4_1: the layer below sits behind a pool (no absorbed ReLU backward)
    (1, 81, 81, 256, 256, 1, True),       # conv3_2 / conv3_3
    (2, 20, 30, 256, 512, 1, True),       # a ragged last tile on both sides
    (2, 81, 81, 128, 256, 1, False),      # conv3_1: a 128-channel x (two taps per weight-gradient tile, half an n-tile in the data gradient)
    (2, 41, 41, 512, 1024, 12, True),     # a dilated layer: the entry point runs the two launches one after the other
])
def test_igemm_merged_backward_equals_the_two_launches(ops, B, H, W, cin, cout, dil, absorb):
    """dsrg_conv_igemm_backward_bf16: the data gradient's tiles and the weight gradient's workgroups of one layer in ONE grid
    (conv_igemm_bwd_kernel) — the same device code per block: data gradient (+ the absorbed ReLU backward and bias gradient of
    the layer below) bit-equal to the separate launch, weight gradient equal up to fp32 reassociation (its pixel split is chosen
    for the merged grid), deterministic"""
    x = torch.relu(torch.randn(B, cin, H, W, device="cuda")).bfloat16().contiguous(memory_format=CL)     # the layer's input = a ReLU output
    g = torch.randn(B, cout, H, W, device="cuda").bfloat16().contiguous(memory_format=CL)
    w = (torch.randn(cout, cin, 3, 3, device="cuda") * 0.03)
    pd = ops.pack_conv_weight(w, for_dgrad=True)
    gx, gw, gb = ops.conv_igemm_backward(g, pd, x, dil, x if absorb else None, 2.0 if absorb else 1.0)
    if absorb:
        (gx_ref,), (gb_ref,) = ops.conv_igemm_dgrad([g], [pd], [x], [dil], 3, 2.0)
        assert torch.equal(gb, gb_ref)
    else:
        (gx_ref,) = ops.conv_igemm([g], [pd], [None], [dil], 3, False, stream_k=False)
        assert gb is None
    (gw_ref,) = ops.conv_igemm_wgrad([x], [g], [dil], 3)
    assert torch.equal(gx, gx_ref)
    # the weight-gradient half may cut the pixels finer than the stand-alone launch does (its split is chosen for the merged grid):
    # the same terms in the same order, other partial sums
    assert (gw - gw_ref).abs().max() <= 2e-5 * gw_ref.abs().max()
    gx2, gw2, _ = ops.conv_igemm_backward(g, pd, x, dil, x if absorb else None, 2.0 if absorb else 1.0)
    assert torch.equal(gx, gx2) and torch.equal(gw, gw2)                                                   # deterministic
    # and right: the weight gradient against torch
    wr = torch.zeros(cout, cin, 3, 3, device="cuda", requires_grad=True)
    F.conv2d(x.float(), wr, None, padding=dil, dilation=dil).backward(g.float())
    assert (gw - wr.grad).abs().max() <= 2e-3 * wr.grad.abs().max() + 1e-4


def _unpack(p):
    o, cc, taps, _ = p.shape
    k = int(round(taps ** 0.5))
    return p.float().permute(0, 1, 3, 2).reshape(o, cc * 64, k, k)


def test_igemm_rejects_unsupported_shapes(ops):
    from dsrg_amd._lib import DsrgError
    assert ops.conv_igemm_supported(512, 1024, 3) and ops.conv_igemm_supported(1024, 1024, 1)
    assert ops.conv_igemm_supported(256, 128, 3) and not ops.conv_igemm_supported(512, 64, 3) and not ops.conv_igemm_supported(96, 256, 3)
    assert ops.conv_igemm_wgrad_supported(128, 256, 3) and not ops.conv_igemm_wgrad_supported(128, 256, 1)
    assert not ops.conv_igemm_wgrad_supported(64, 256, 3) and not ops.conv_igemm_wgrad_supported(256, 128, 3)
    assert ops.conv_igemm_wgrad_launchable(64, 256, 1) and ops.conv_igemm_wgrad_launchable(128, 128, 3) and not ops.conv_igemm_wgrad_launchable(96, 256, 1)
    x = torch.zeros(1, 64, 4, 4, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=CL)
    with pytest.raises(ValueError):
        ops.conv_igemm([x], [torch.zeros(256, 1, 9, 64, device="cuda")], [None], [1], 3, False)      # float32 kernel
    # (a 64-channel output is launchable — half-empty tiles, the ResNet res2 1x1 layers — though not the recommended route: supported() says no)
    (y64,) = ops.conv_igemm([x], [torch.zeros(64, 1, 9, 64, device="cuda", dtype=torch.bfloat16)], [None], [1], 3, False)
    assert tuple(y64.shape) == (1, 64, 4, 4) and not y64.float().abs().any()
    with pytest.raises(DsrgError):
        ops.conv_igemm([x], [torch.zeros(192, 1, 9, 64, device="cuda", dtype=torch.bfloat16)], [None], [1], 3, False)


@pytest.mark.parametrize("B,H,W,cin,cout,k,dil", [
    (1, 5, 7, 256, 256, 3, 1),           # one K-step, every border case
    (2, 19, 23, 256, 256, 3, 2),         # several steps, ragged split, map narrower than a step (64 pixels span 2-3 rows)
    (1, 41, 41, 512, 512, 3, 2),         # conv5_x
    (2, 41, 41, 512, 1024, 3, 12),       # fc6_2
    (2, 41, 41, 1024, 1024, 1, 1),       # fc7
    (1, 81, 81, 256, 256, 3, 1),         # conv3_2
    (3, 1, 1, 256, 256, 3, 1),
    (1, 3, 100, 256, 512, 3, 24),        # map wider than a step; dilation beyond the height
    (2, 27, 31, 128, 256, 3, 1),         # conv3_1: two taps of 128 channels per column tile, the tenth "tap" of the last tile empty
    (1, 81, 81, 128, 256, 3, 1),
    (2, 37, 35, 256, 64, 1, 1),          # ResNet res2 / res3 (round 6): partly empty tiles — 64 outputs
    (2, 37, 35, 64, 256, 1, 1),          #   64 inputs
    (1, 50, 41, 64, 64, 1, 1),
    (2, 33, 33, 128, 512, 1, 1),         #   128 inputs, 1x1 (not the two-tap form)
    (2, 33, 33, 512, 128, 1, 1),
    (2, 33, 35, 128, 128, 3, 1),         #   two-tap tiles, 128 outputs
    (1, 29, 31, 64, 128, 3, 2),          #   a 3x3 kernel over 64 inputs
    (1, 21, 23, 320, 192, 1, 1),         #   the last tile of each axis partly empty
])
def test_igemm_weight_gradient_matches_torch(ops, B, H, W, cin, cout, k, dil):
    x, w, _ = _case(B, H, W, cin, cout, k, 9)
    g = torch.randn(B, cout, H, W, device="cuda").bfloat16().contiguous(memory_format=CL)
    wr = w.float().requires_grad_(True)
    F.conv2d(x.float(), wr, None, padding=dil * (k // 2), dilation=dil).backward(g.float())
    (gw,) = ops.conv_igemm_wgrad([x], [g], [dil], k)
    assert gw.shape == wr.shape and gw.dtype == torch.float32 and gw.is_contiguous(memory_format=CL)
    err = (gw - wr.grad).abs().max()
    assert err <= 2e-3 * wr.grad.abs().max() + 1e-4, (float(err), float(wr.grad.abs().max()))       # fp32 sums of exact products
    (again,) = ops.conv_igemm_wgrad([x], [g], [dil], k)
    assert torch.equal(gw, again)                                                    # deterministic
    (gb,) = ops.conv_igemm_wgrad([x], [g], [dil], k, out_dtype=torch.bfloat16)
    assert torch.equal(gb, gw.bfloat16())


def test_igemm_weight_gradient_four_branches(ops):
    B, H, W, cin, cout = 1, 41, 41, 512, 1024
    x, _, _ = _case(B, H, W, cin, cout, 3, 31)
    gs = [torch.randn(B, cout, H, W, device="cuda").bfloat16().contiguous(memory_format=CL) for _ in range(4)]
    dils = [6, 12, 18, 24]
    together = ops.conv_igemm_wgrad([x] * 4, gs, dils, 3)
    for i in range(4):
        (alone,) = ops.conv_igemm_wgrad([x], [gs[i]], [dils[i]], 3)
        assert (together[i] - alone).abs().max() <= 1e-4 * alone.abs().max()         # another pixel split: another summation order
    wr = torch.zeros(cout, cin, 3, 3, device="cuda", requires_grad=True)
    F.conv2d(x.float(), wr, None, padding=18, dilation=18).backward(gs[2].float())
    assert (together[2] - wr.grad).abs().max() <= 2e-3 * wr.grad.abs().max() + 1e-4


def test_backbone_igemm_route_is_as_close_to_float32_as_the_im2col_route():
    """VGG16-ASPP forward + backward (dropout off, same weights, input and score gradient): the implicit-GEMM route (conv3_x ..
    fc6_k, fc7_k weight gradients) and the im2col + hipBLASLt route of the earlier rounds under bf16 autocast, each against the
    float32 backbone.  bf16 activations make every parameter gradient noisy, the more the deeper the backward chain (a
    relative distance of 2e-3 at fc8 grows to 0.8 at conv1_1 with default initialisation: tools/igemm_route_diag.py), so
    the routes are not compared with each other: the new one must be no further from float32 than the old one."""
    from dsrg_amd import backbone
    torch.manual_seed(3)
    net = backbone.VGG16ASPP(dropout=0.0).cuda().to(memory_format=CL)
    x = torch.randn(2, 3, 161, 161, device="cuda").contiguous(memory_format=CL)
    gout, res = None, {}
    try:
        for tag, route, amp in (("igemm", True, True), ("im2col", False, True), ("fp32", False, False)):
            backbone._IGEMM = route
            net.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                y = net(x)
            gout = torch.randn_like(y) if gout is None else gout
            y.backward(gout)
            res[tag] = (y.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters()})
    finally:
        backbone._IGEMM = True
    rel = lambda a, b: float((a.float() - b).norm() / b.norm().clamp_min(1e-20))      # noqa: E731
    assert res["igemm"][0].dtype == torch.float32
    assert rel(res["igemm"][0], res["fp32"][0]) <= 1.25 * rel(res["im2col"][0], res["fp32"][0]) + 1e-3
    worse = []
    for n, ref in res["fp32"][1].items():
        a, b = res["igemm"][1][n], res["im2col"][1][n]
        assert a.dtype == torch.float32 and a.shape == ref.shape
        if rel(a, ref) > 1.3 * rel(b, ref) + 0.01:
            worse.append((n, rel(a, ref), rel(b, ref)))
    assert not worse, worse


def test_pack_kernel_matches_the_torch_copies(ops):
    """dsrg_pack_conv_weight_f32: both packed forms in one pass == pack_conv_weight's strided torch copies, bit for bit"""
    for cout, cin, k in [(256, 128, 3), (512, 512, 3), (1024, 512, 3), (1024, 1024, 1), (64, 64, 3)]:
        w = torch.randn(cout, cin, k, k, device="cuda").contiguous(memory_format=CL)
        fwd, dg = ops.pack_conv_weight_pair(w)
        assert torch.equal(fwd, ops.pack_conv_weight(w)) and torch.equal(dg, ops.pack_conv_weight(w, for_dgrad=True))
        only_f, none_d = ops.pack_conv_weight_pair(w, True, False)
        assert none_d is None and torch.equal(only_f, fwd)


def test_igemm_fused_dropout(ops):
    """Dropout in the convolution's epilogue: every element is 0 or relu(conv) / (1 - p) (p in steps of 1/256), the kept
    share is 1 - p, the mask is a function of (seed, branch, position) only, and 3x3 and 1x1 launches both carry it"""
    for k, cin, cout, dil in [(3, 128, 512, 2), (1, 256, 256, 1)]:
        x, w, b = _case(2, 41, 41, cin, cout, k, 40 + k)
        packed = ops.pack_conv_weight(w)
        (plain,) = ops.conv_igemm([x], [packed], [b], [dil], k, True)
        for p in (0.5, 0.25):
            (a1,) = ops.conv_igemm([x], [packed], [b], [dil], k, True, p, 1234)
            (a2,) = ops.conv_igemm([x], [packed], [b], [dil], k, True, p, 1234)
            (a3,) = ops.conv_igemm([x], [packed], [b], [dil], k, True, p, 1235)
            assert torch.equal(a1, a2) and not torch.equal(a1, a3)
            kept = a1 != 0
            pos = plain > 0
            assert not (kept & ~pos).any()                                           # nothing appears where the ReLU is off
            share = float((kept & pos).sum()) / float(pos.sum())
            assert abs(share - (1 - p)) < 0.01, share
            want = (plain.float() / (1 - p)).bfloat16()                              # one rounding from the fp32 accumulator in both
            diff = (a1.float() - want.float())[kept].abs()
            assert (diff <= 0.008 * want.float()[kept].abs() + 1e-6).all()           # the plain output was itself rounded once
            # masks of different branches differ
            two = ops.conv_igemm([x, x], [packed, packed], [b, b], [dil, dil], k, True, p, 1234)
            assert torch.equal(two[0], a1) and not torch.equal(two[1], a1)
            both = ((two[0] != 0) & (two[1] != 0) & pos).sum().item() / pos.sum().item()
            assert abs(both - (1 - p) ** 2) < 0.01                                   # independent


@pytest.mark.parametrize("B,H,W,cin,cout,k,dils", [
    (16, 41, 41, 512, 512, 3, [2]),                  # 212 tiles of 72 steps on 256 CUs: every workgroup 59.6 steps
    (16, 41, 41, 512, 256, 3, [1]),                  # 106 tiles: tiles cut in three
    (4, 41, 41, 512, 1024, 3, [6, 12, 18, 24]),      # four branches, 7 x 4 x 4 tiles
    (4, 81, 81, 256, 128, 3, [1]),                   # half-width n-tiles (waves 4-7 idle) through the stream-K form
    (16, 41, 41, 1024, 1024, 1, [1]),                # 1x1: 16 steps per tile
])
def test_igemm_stream_k_equals_whole_tiles(ops, B, H, W, cin, cout, k, dils):
    """the stream-K form (K-steps of all tiles dealt out evenly, cut tiles completed through the scratch) against the one-tile-
    per-workgroup launch: equal up to the fp32 summation order of a cut tile, bit-reproducible, same Dropout mask, no
    workgroup gave up waiting"""
    n = len(dils)
    xs, ws, bs = [], [], []
    for i in range(n):
        x, w, b = _case(B, H, W, cin, cout, k, 60 + i)
        xs.append(x); ws.append(ops.pack_conv_weight(w)); bs.append(b)
    for p in (0.0, 0.5):
        whole = ops.conv_igemm(xs, ws, bs, dils, k, True, p, 77, stream_k=False)
        ops.set_igemm_variant(4)                     # stream-K wherever legal (by default only where it was measured to win)
        try:
            cut = ops.conv_igemm(xs, ws, bs, dils, k, True, p, 77, stream_k=True)
            assert ops.conv_igemm_stream_k_status() == 0
            again = ops.conv_igemm(xs, ws, bs, dils, k, True, p, 77, stream_k=True)
        finally:
            ops.set_igemm_variant(-1)
        for a, b_, c in zip(whole, cut, again):
            assert torch.equal(b_, c)
            assert torch.equal(a == 0, b_ == 0) or ((a == 0) != (b_ == 0)).float().mean() < 1e-4     # a sum on the edge of the ReLU
            assert (a.float() - b_.float()).abs().max() <= 0.01 * a.float().abs().max() + 1e-3


@pytest.mark.parametrize("B,H,W,cf,cb,k,dils,scale", [
    (2, 41, 41, 512, 512, 3, [1], 1.0),            # conv4_2 -> conv4_1's ReLU
    (1, 33, 29, 256, 256, 3, [2], 1.0),            # ragged pixel tail, dilation
    (2, 41, 41, 1024, 1024, 1, [1, 1, 1, 1], 2.0),  # fc7_k -> fc6_k's ReLU + Dropout (scale 2), four branches
    (1, 20, 23, 512, 256, 3, [1, 3], 2.0),
])
def test_igemm_dgrad_absorbs_relu_backward_and_bias_gradient(ops, B, H, W, cf, cb, k, dils, scale):
    """dsrg_conv_igemm_dgrad_bf16 == the plain data gradient followed by the ReLU / Dropout backward of the layer below
    (values kept where that layer's output is positive, times the Dropout scale) and that layer's bias gradient (column sums
    of what was stored).  cf: channels of the incoming gradient (the layer's outputs), cb: channels of the layer below."""
    n = len(dils)
    torch.manual_seed(77)
    gs = [torch.randn(B, cf, H, W, device="cuda").bfloat16().contiguous(memory_format=CL) for _ in range(n)]
    ws = [(torch.randn(cf, cb, k, k, device="cuda") * (2.0 / (cf * k * k)) ** 0.5) for _ in range(n)]
    # outputs of the layer below: ReLU output with zeros, a few negative zeros / negatives must count as "off" too
    ys = []
    for _ in range(n):
        y = torch.relu(torch.randn(B, cb, H, W, device="cuda")).bfloat16().contiguous(memory_format=CL)
        y.permute(0, 2, 3, 1).view(-1)[::97] = -0.0
        ys.append(y)
    packs = [ops.pack_conv_weight(w, for_dgrad=True) for w in ws]
    plain = ops.conv_igemm(gs, packs, None, dils, k, False, stream_k=False)
    got, gb = ops.conv_igemm_dgrad(gs, packs, ys, dils, k, scale)
    for g in range(n):
        want = torch.where(ys[g] > 0, (plain[g].float() * scale).bfloat16(), torch.zeros_like(plain[g]))
        assert got[g].is_contiguous(memory_format=CL)
        if scale in (1.0, 2.0):                                          # a power of two: the same roundings
            assert torch.equal(got[g], want)
        else:
            _close(got[g], want.float(), "masked data gradient")
        ref_b = want.float().sum((0, 2, 3))
        assert gb[g].dtype == torch.float32 and gb[g].shape == (cb,)
        assert float((gb[g] - ref_b).abs().max()) <= 2e-3 * float(ref_b.abs().max()) + 1e-4
    only, none = ops.conv_igemm_dgrad(gs, packs, ys, dils, k, scale, bias_grad=False)
    assert none is None and all(torch.equal(a, b) for a, b in zip(only, got))


def _grads(net, x, gout, amp=True, seed=11):
    net.zero_grad(set_to_none=True)
    torch.manual_seed(seed)                                              # the fused Dropout's seeds come from torch's CPU generator
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        y = net(x)
    gout = torch.randn_like(y) if gout is None else gout
    y.backward(gout)
    return y.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters()}, gout


@pytest.mark.parametrize("widths,size", [((256, 256, 512, 256), 41), ((64, 64, 64), 97), ((64, 128, 128), 50), ((128, 128, 64, 64), 33)])
def test_conv_chain_fusion_is_bit_identical_for_3x3_chains(widths, size):
    """conv -> ReLU -> conv -> ReLU -> conv (GemmConv2d(chain_input=True)): the upper layer's data gradient with the lower
    layer's ReLU backward and bias gradient folded in (backbone._GradLink) against the separate relu_bwd_bias passes: masked
    gradients are the same bits, so weight gradients are equal and bias gradients agree to the fp32 summation order"""
    from dsrg_amd import backbone
    torch.manual_seed(5)
    G = backbone.GemmConv2d
    wide = widths[0] >= 256                                              # implicit-GEMM route (dilation on the middle layer) / direct kernels
    net = torch.nn.Sequential(*[G(a, b, 3, padding=2 if (wide and i == 1) else 1, dilation=2 if (wide and i == 1) else 1, fuse_relu=True,
                                  chain_input=i > 0) for i, (a, b) in enumerate(zip(widths[:-1], widths[1:]))]).cuda().to(memory_format=CL)
    x = torch.randn(4, widths[0], size, size, device="cuda").contiguous(memory_format=CL).requires_grad_(True)
    res, gout = {}, None
    try:
        for tag, on in (("fused", True), ("separate", False)):
            backbone._FUSE_CHAIN = on
            x.grad = None
            y, gr, gout = _grads(net, x, gout)
            res[tag] = (y, gr, x.grad.clone())
    finally:
        backbone._FUSE_CHAIN = True
    assert torch.equal(res["fused"][0], res["separate"][0]) and torch.equal(res["fused"][2], res["separate"][2])
    for n, ref in res["separate"][1].items():
        a = res["fused"][1][n]
        if n.endswith("bias"):
            assert float((a - ref).abs().max()) <= 1e-3 * float(ref.abs().max()) + 1e-6, n
        else:
            assert torch.equal(a, ref), n


def test_backbone_chain_fusion_matches_the_separate_backward_passes():
    """VGG16-ASPP with Dropout on, same seeds: chains fused (conv3_x .. conv5_x, fc6_k -> fc7_k through the 1x1 implicit-GEMM
    data gradient instead of hipBLASLt's) against the separate passes.  fc7 / fc8 gradients do not see the difference; below
    them only the fp32 summation order of fc7's data gradient differs (bf16 roundings flip here and there)"""
    from dsrg_amd import backbone
    torch.manual_seed(5)
    net = backbone.VGG16ASPP(dropout=0.5).cuda().to(memory_format=CL).train()
    x = torch.randn(4, 3, 161, 161, device="cuda").contiguous(memory_format=CL)
    res, gout = {}, None
    try:
        for tag, on in (("fused", True), ("separate", False)):
            backbone._FUSE_CHAIN = on
            y, gr, gout = _grads(net, x, gout)
            res[tag] = (y, gr)
    finally:
        backbone._FUSE_CHAIN = True
    assert torch.equal(res["fused"][0], res["separate"][0])
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-20))      # noqa: E731
    for n, ref in res["separate"][1].items():
        a = res["fused"][1][n]
        top = n.startswith("branches") and (".3." in n or ".6." in n)          # fc7_k, fc8_k
        if top and not n.endswith("bias"):
            assert torch.equal(a, ref), n
        else:
            assert rel(a, ref) < (1e-3 if top else 0.03), (n, rel(a, ref))


@pytest.mark.parametrize("width", [64, 256])
def test_chain_link_declines_when_the_output_has_a_second_consumer(width):
    """GemmConv2d(chain_input=True) is a promise that the input feeds nothing else.  If it does after all, autograd hands the lower
    node the SUM of the gradients — not the tensor the upper node left — and the node runs its own ReLU backward on it (right on an
    already masked part, right on the rest): same gradients as with the separate passes"""
    from dsrg_amd import backbone
    torch.manual_seed(8)
    G = backbone.GemmConv2d
    a = G(width, width, 3, padding=1, fuse_relu=True).cuda().to(memory_format=CL)
    b = G(width, width, 3, padding=1, fuse_relu=True, chain_input=True).cuda().to(memory_format=CL)
    x = torch.randn(2, width, 41, 41, device="cuda").contiguous(memory_format=CL)
    res = {}
    try:
        for tag, on in (("fused", True), ("separate", False)):
            backbone._FUSE_CHAIN = on
            a.zero_grad(set_to_none=True); b.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y1 = a(x)
                y2 = b(y1)
            (y2.float().sum() + 0.5 * y1.float().square().sum()).backward()
            res[tag] = [p.grad.clone() for p in list(a.parameters()) + list(b.parameters())]
    finally:
        backbone._FUSE_CHAIN = True
    for u, v in zip(res["fused"], res["separate"]):
        assert float((u - v).norm()) <= 2e-3 * float(v.norm()) + 1e-6
