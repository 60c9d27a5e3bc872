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
import torch.nn.functional as F


import os as _os
# output channels from which the weight gradient of a 3x3 convolution at the 41x41 stages is the GEMM im2col(x)^T @ g
# (tools/wgrad_ab.sh measures the alternatives)
_WGRAD_MIN_COUT = int(_os.environ.get("DSRG_WGRAD_MIN_COUT", "512"))
# conv1_2 / conv2_1 / conv2_2 by the direct MFMA kernel: "1" all of them, "64" only conv1_2, "0" none (MIOpen / im2col + GEMM)
_DIRECT_CONV = _os.environ.get("DSRG_DIRECT_CONV", "1")
_DIRECT_C3 = _os.environ.get("DSRG_DIRECT_C3", "1") == "1"          # conv1_1 (3 -> 64) forward with bias + ReLU in one pass (0: MIOpen + 2 passes)
_DIRECT_FN = _os.environ.get("DSRG_DIRECT_FN", "1") == "1"         # conv1_1 … conv2_2 take the float32 master parameters (_DirectConvFn; 0: through autocast's casts)
_FUSE_POOL = _os.environ.get("DSRG_FUSE_POOL", "1") == "1"           # pool1-3 inside the conv node: pool backward + ReLU mask + bias gradient in one pass
_GEMM_1X1_BWD = _os.environ.get("DSRG_GEMM_1X1_BWD", "1") == "1"   # 1x1 layers (fc7): both gradients as hipBLASLt GEMMs (0: MIOpen/CK)
_DIRECT_WGRAD = _os.environ.get("DSRG_DIRECT_WGRAD", "1") == "1"   # their weight gradients by the direct kernel too (0: MIOpen)
_GEMM_DGRAD_MAX_MAP = int(_os.environ.get("DSRG_GEMM_DGRAD_MAX_MAP", "2048"))   # largest map (pixels) whose 3x3 data gradient is im2col(g) + GEMM
_WGRAD_T = _os.environ.get("DSRG_WGRAD_T", "1") == "1"     # g^T @ im2col(x) (1) or im2col(x)^T @ g (0): same numbers, other solution
# 3x3 layers with >= 256 output channels (conv3_x, conv4_x, conv5_x, fc6_k) through the implicit-GEMM kernels (csrc/conv_igemm.hip:
# no im2col matrix in the forward, the data gradient or the weight gradient; the four fc6_k in one launch); "0": the im2col +
# hipBLASLt route of rounds 1-3
_IGEMM = _os.environ.get("DSRG_IGEMM", "1") == "1"
_FC7_IGEMM = _os.environ.get("DSRG_FC7_IGEMM", "1") == "1"   # fc7_k forward by the 1x1 implicit-GEMM launch when Dropout follows (mask fused)


def _im2col_gemm(x, weight, bias, dilation, relu, want_cols=False):
    """(B,C,H,W) bf16 -> conv(+ReLU) output as a channels_last tensor: NHWC im2col (HIP) + one hipBLASLt GEMM whose
    epilogue adds the bias (and applies the ReLU)."""
    B, C, H, W = x.shape
    cout, k = weight.shape[0], weight.shape[2]
    xn = x.permute(0, 2, 3, 1)                                       # free for channels_last tensors
    if not xn.is_contiguous():
        xn = xn.contiguous()
    if k == 1:
        a = xn.reshape(-1, C)
    elif (C * x.element_size()) % 16 == 0 and x.element_size() in (2, 4):
        from .ops import im2col3x3_nhwc                               # one bandwidth-bound HIP kernel (16-byte channel groups)
        a = im2col3x3_nhwc(xn, dilation)
    else:
        p = dilation
        xp = F.pad(xn, (0, 0, p, p, p, p))
        a = torch.cat([xp[:, dy * p:dy * p + H, dx * p:dx * p + W, :] for dy in range(3) for dx in range(3)],
                      dim=-1).reshape(-1, 9 * C)
    wmat = weight.permute(2, 3, 1, 0).reshape(k * k * C, cout)
    # write straight into a channels_last (B,cout,H,W) tensor: its NHWC memory is the GEMM's C matrix
    out = torch.empty((B, cout, H, W), dtype=a.dtype, device=a.device, memory_format=torch.channels_last)
    o2 = out.permute(0, 2, 3, 1).view(-1, cout)
    if bias is None:
        torch.mm(a, wmat, out=o2)
    elif relu:
        torch._addmm_activation(bias, a, wmat, out=o2)
    else:
        torch.addmm(bias, a, wmat, out=o2)
    return (out, a) if want_cols else out


class _ConvFn(torch.autograd.Function):
    """3x3 / 1x1 stride-1 'same' convolution (+ ReLU (+ Dropout)).  Forward: explicit NHWC im2col + one hipBLASLt GEMM
    when `gemm` (at the 41x41 stages, 76 % of the backbone flops, MIOpen's forward kernels reach ~180 TFLOP/s on MI355X,
    the GEMM route 300-1200: tools/conv_probe.py), MIOpen otherwise.  Backward: the ReLU mask, the dropout mask and scale
    and the bias gradient come from one fused HIP pass (ops.relu_bwd_bias: the output of ReLU + Dropout is positive
    exactly where both masks pass); data and weight gradients go to hipBLASLt GEMMs where that beats MIOpen/CK (the 41x41
    stages, see the comments in backward) and to MIOpen/CK otherwise."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.bfloat16)
    def forward(ctx, x, weight, bias, dilation, relu, gemm, drop_p, pool=None, link_in=None, link_out=None):
        k = weight.shape[2]
        cols = None
        # 64 / 128 channels on both sides (conv1_2 at full resolution, conv2_1 / conv2_2 at half): the direct MFMA kernel
        # (weights in registers, bias + ReLU in the epilogue; csrc/conv_direct.hip) — MIOpen's implicit GEMM runs conv1_2 at
        # ~255 TFLOP/s, the im2col route would move 1.9 GB there and 0.96 GB for conv2_2
        shape = tuple(weight.shape[:2])
        direct = k == 3 and dilation == 1 and x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and (
            (_DIRECT_CONV == "1" and shape[0] in (64, 128) and shape[1] in (64, 128)) or (_DIRECT_CONV == "64" and shape == (64, 64))
            or (_DIRECT_CONV == "1" and _DIRECT_C3 and shape == (64, 3)))
        if direct:
            from .ops import conv3x3_direct
            out = conv3x3_direct(x, weight, bias, relu)
        elif gemm:
            # the im2col matrix is kept for the layers whose weight gradient is a GEMM too (see backward)
            keep = k == 3 and weight.shape[0] >= _WGRAD_MIN_COUT and x.shape[1] % 8 == 0 and x.shape[2] * x.shape[3] <= 2048
            out = _im2col_gemm(x, weight, bias, dilation, relu, want_cols=keep)
            if keep:
                out, cols = out
        else:
            out = F.conv2d(x.contiguous(memory_format=torch.channels_last), weight, bias, 1, dilation * (k // 2), dilation)
            if relu:
                out.relu_()
        if drop_p > 0.0:
            out = torch.ops.aten.native_dropout(out, drop_p, True)[0]    # out = relu * mask / (1 - p)
        code, pooled = None, None
        if pool is not None:
            # conv + ReLU + 3x3 max pool as one autograd node (conv1_2, conv2_2, conv3_3): the pool's backward then masks with
            # this ReLU and sums the bias gradient in the same pass (ops.maxpool3x3_bwd_relu) instead of handing an unmasked
            # gradient to a separate relu_bwd_bias pass — three fewer passes over the largest activations of the net
            from .ops import maxpool3x3_fwd
            # (relu_input: the window codes carry this ReLU's mask, so the backward never reads the full-resolution output again
            # and the node does not keep it)
            pooled, code = maxpool3x3_fwd(out, pool[0], pool[1], relu_input=True)
        ctx.save_for_backward(x, weight, out if (relu and pool is None) else None, cols, code)
        ctx.out_shape = tuple(out.shape)
        ctx.dilation, ctx.k, ctx.relu, ctx.scale, ctx.gemm = dilation, k, relu, 1.0 / (1.0 - drop_p), gemm
        ctx.direct, ctx.pool = direct, pool
        # _GradLink (see below): link_in — x is the ReLU output of the node in front and feeds nothing else; link_out — ours
        ctx.link_in = link_in if (direct and link_in is not None and link_in.scale == 1.0 and x.shape[1] in (64, 128) and
                                  x.is_contiguous(memory_format=torch.channels_last)) else None
        ctx.link_out = link_out if (relu and pool is None and drop_p == 0.0) else None
        if ctx.link_out is not None:
            ctx.link_out.scale, ctx.link_out.gb = 1.0, None
        return out if pool is None else pooled

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        x, weight, y, cols, code = ctx.saved_tensors
        pad = ctx.dilation * (ctx.k // 2)
        cout = weight.shape[0]
        fused = g.dtype == torch.bfloat16 and ((cout % 8 == 0 and cout <= 2048) or (not ctx.relu and cout <= 256))
        gb_left = ctx.link_out.take(g) if ctx.link_out is not None else None
        if gb_left is not None:
            # the consumer's data gradient came masked by this node's ReLU, with the bias gradient beside it
            gb = gb_left
            fused = True
        elif ctx.pool is not None:                                      # forward guaranteed bf16, ReLU, no dropout, cout | 2048
            from .ops import maxpool3x3_bwd_relu
            g, gb = maxpool3x3_bwd_relu(g, code, ctx.out_shape, ctx.pool[0])
            fused = True
        elif fused and ctx.relu:
            from .ops import relu_bwd_bias
            g, gb = relu_bwd_bias(g, y, ctx.scale)
        elif fused:
            from .ops import bias_grad                                  # e.g. the 21-channel fc8 outputs
            g = g.contiguous(memory_format=torch.channels_last)
            gb = bias_grad(g)
        else:
            if ctx.relu:
                g = g * (y > 0) * ctx.scale
            g = g.contiguous(memory_format=torch.channels_last)
        # data gradient of a 3x3 'same' convolution = the forward convolution of g with the flipped, transposed kernel:
        # the im2col + hipBLASLt route again (~1.2 PFLOP/s at the 41x41 stages against 550-630 TFLOP/s for CK's dgrad)
        gemm_dgrad = ctx.gemm and ctx.k == 3 and ctx.needs_input_grad[0] and g.dtype in (torch.bfloat16, torch.float32) \
            and cout % 8 == 0 and x.shape[2] * x.shape[3] <= _GEMM_DGRAD_MAX_MAP    # larger maps: the im2col of g costs more than it saves (81x81, again with nontemporal im2col stores: 1 221 / 1 225 against 1 233 / 1 227 images/s)
        gx = None
        gemm_1x1 = _GEMM_1X1_BWD and ctx.gemm and ctx.k == 1 and g.dtype in (torch.bfloat16, torch.float32) and cout % 8 == 0 \
            and x.shape[1] % 8 == 0 and g.dtype == x.dtype
        if gemm_1x1:
            # fc7 (1024 -> 1024, 1x1): both gradients are plain GEMMs over the NHWC matrices (MIOpen's wrw 120 us and CK's dgrad
            # 98 us per branch at 56 GFLOP each)
            cin = x.shape[1]
            g2d = g.permute(0, 2, 3, 1).reshape(-1, cout)
            if ctx.needs_input_grad[0]:
                gx = torch.mm(g2d, weight.reshape(cout, cin)).view(x.shape[0], x.shape[2], x.shape[3], cin).permute(0, 3, 1, 2)
            gemm_dgrad = True
        elif ctx.direct and ctx.needs_input_grad[0] and g.dtype == torch.bfloat16 and x.shape[1] in (64, 128):
            # the data gradient is the same convolution with the kernel flipped and its channel axes swapped
            from .ops import conv3x3_direct, conv3x3_direct_dgrad
            if _FUSE_CHAIN and ctx.link_in is not None:
                gx, gb_below = conv3x3_direct_dgrad(g, weight.flip(2, 3).transpose(0, 1), x)
                ctx.link_in.leave(gx, gb_below)
            else:
                gx = conv3x3_direct(g, weight.flip(2, 3).transpose(0, 1), None, False)
            gemm_dgrad = True
        elif gemm_dgrad and cout > x.shape[1] and x.shape[1] % 8 == 0 and g.dtype == torch.bfloat16:
            # more output than input channels (fc6: 1024 vs 512): g @ W^T first, then gather the nine taps (col2im) —
            # half the traffic of an im2col of the wide g
            from .ops import col2im3x3_nhwc
            B_, cin, H_, W_ = x.shape
            wmat = weight.permute(2, 3, 1, 0).reshape(9 * cin, cout)
            gx = col2im3x3_nhwc(torch.mm(g.permute(0, 2, 3, 1).reshape(-1, cout), wmat.t()), B_, H_, W_, cin, ctx.dilation)
        elif gemm_dgrad:
            gx = _im2col_gemm(g, weight.flip(2, 3).transpose(0, 1), None, ctx.dilation, False)
        # weight gradient = im2col(x)^T @ g, again one hipBLASLt GEMM (K = B*H*W), for the 41x41 layers with >= 512 output
        # channels: the 512 -> 1024 dilated fc6 layers (MIOpen's wrw 650 TFLOP/s there; 883 -> 902 images/s in round 1) and,
        # since the hipBLASLt solutions are picked by TunableOp, the 512 -> 512 / 256 -> 512 layers too (round 2, A/B on one
        # box: 932.8 / 931.4 images/s with the 1024 threshold, 939.9 / 938.9 with 512, 934.1 / 932.4 with 256; the transposed
        # product g^T @ im2col(x) another +0.6 %)
        gemm_wgrad = gemm_dgrad and x.shape[2] * x.shape[3] <= 2048 and x.shape[1] % 8 == 0 and cout >= _WGRAD_MIN_COUT
        gw = None
        if gemm_1x1:
            x2d = x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])       # NHWC memory of a channels_last activation
            gw = torch.mm(g2d.t(), x2d).view(cout, x.shape[1], 1, 1)
            gemm_wgrad = True
        elif ctx.direct and _DIRECT_WGRAD and g.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and \
                (x.shape[1], cout) in ((3, 64), (64, 64), (64, 128), (128, 128)):
            # the narrow full-resolution layers again: MIOpen's wrw kernels run them at ~250 TFLOP/s (conv1_2: 0.48 ms)
            from .ops import conv3x3_wgrad
            gw = conv3x3_wgrad(x, g)
            gemm_wgrad = True
        elif gemm_wgrad:
            cin = x.shape[1]
            if cols is None or cols.dtype != g.dtype:
                from .ops import im2col3x3_nhwc
                cols = im2col3x3_nhwc(x.permute(0, 2, 3, 1).contiguous(), ctx.dilation)  # (M, 9*Cin)
            g2d = g.permute(0, 2, 3, 1).reshape(-1, cout)                                # (M, Cout), NHWC memory
            if _WGRAD_T:
                gw = torch.mm(g2d.t(), cols).view(cout, 3, 3, cin).permute(0, 3, 1, 2)
            else:
                gw = torch.mm(cols.t(), g2d).view(3, 3, cin, cout).permute(3, 2, 0, 1)
        mask = [ctx.needs_input_grad[0] and not gemm_dgrad, not gemm_wgrad, not fused]
        gx2 = gw2 = gb2 = None
        if any(mask):
            gx2, gw2, gb2 = torch.ops.aten.convolution_backward(
                g, x, weight, None if fused else [weight.shape[0]], [1, 1], [pad, pad],
                [ctx.dilation, ctx.dilation], False, [0, 0], 1, mask)
        return (gx if gemm_dgrad else gx2), (gw if gemm_wgrad else gw2), (gb if fused else gb2), None, None, None, None, None, None, None


class _GradLink:
    """side channel between two adjacent nodes of a conv chain: when the upper node's data gradient was taken with the lower
    node's ReLU (+ Dropout) backward and bias gradient folded into its store (ops.conv_igemm_dgrad), it leaves the bias gradient
    here and the lower node's backward takes the incoming gradient as already masked.  Only wired where the model declares the
    lower node's output to have this one consumer (GemmConv2d(chain_input=True); VGG16ASPP's fc6 -> fc7)."""
    __slots__ = ("scale", "gb", "ptr", "version")

    def __init__(self):
        self.scale, self.gb, self.ptr, self.version = 1.0, None, 0, -1

    def leave(self, gx, gb):
        """upper node: gx is the masked (and scaled) data gradient it returns for the lower node's output, gb that node's bias gradient"""
        self.gb, self.ptr, self.version = gb, gx.data_ptr(), gx._version

    def take(self, g):
        """lower node: the bias gradient if g is exactly the tensor the upper node left (autograd hands a lone gradient through
        untouched; had the output a second consumer after all, the sum would be another tensor or a later version of this one),
        else None — the caller then runs its own ReLU backward, which is still right on a masked gradient unless a Dropout
        scale was applied with it"""
        gb, self.gb = self.gb, None
        if gb is None:
            return None
        if g.data_ptr() == self.ptr and g._version == self.version:
            return gb
        if self.scale != 1.0:
            raise RuntimeError("chain_input: the output of a conv + ReLU + Dropout node declared to have one consumer has several")
        return None


_FUSE_CHAIN = _os.environ.get("DSRG_FUSE_CHAIN", "1") != "0"
_MERGED_BWD = _os.environ.get("DSRG_MERGED_BWD", "1") != "0"     # data + weight gradient of a single-group 3x3 layer in one launch
# (measured and dropped in round 5: packing every wide layer's kernels on a second stream at the start of the step, under the
# HBM-bound conv1_x / conv2_x — 1 822 -> 1 798 images/s: the packs then compete with those kernels for the same HBM)
# (measured and dropped in round 5: the weight gradient of an implicit-GEMM layer on a second stream beside its data gradient,
# to fill the sixth of the chip a 212-tile data-gradient launch leaves idle — 1 761 -> 1 715 images/s, the two 139 KB-LDS kernels
# only take CUs from each other; profiles/r05_wgrad_side_stream_ab.txt)


def _landed(g, slot):
    """the gradient a backward returns for a parameter: when the kernel wrote it into the reducer's slot, a FRESH alias of the slot
    (autograd adopts a gradient as `.grad` without a copy only if nobody else holds the tensor object)"""
    return g.detach() if (slot is not None and g is slot) else g


class _IgemmConvFn(torch.autograd.Function):
    """n convolutions of one geometry (n = 1, or the four ASPP branches) + bias (+ ReLU (+ Dropout) (+ the stride-2 max pool)):
    apply(k, dils, relu, drop_p, pool, n, links_in, links_out, x_1..x_n, w_1..w_n, b_1..b_n) (links: lists of _GradLink or None —
    links_in[i] says x_i is the sole-consumer output of a ReLU node whose backward this node's data gradient absorbs).  k = 3 (conv3_x .. fc6_k): forward, data and weight
    gradient by the implicit-GEMM kernels; k = 1 (fc7_k): forward and data gradient are plain hipBLASLt GEMMs over the NHWC
    matrices (nothing to gather), the weight gradients of all branches one implicit-GEMM launch.
    x: bf16 channels_last; w, b: the float32 master parameters — the kernel is cast to bf16 by the same copy that packs it
    for the kernel, and the weight gradient comes back in float32, so autocast's per-layer casts both ways disappear.
    Backward: ReLU / dropout mask + bias gradient in one fused pass (as _ConvFn), data gradients of all branches in one launch
    (the same kernel on the flipped, transposed kernels), weight gradients in one launch."""

    @staticmethod
    def forward(ctx, k, dils, relu, drop_p, pool, n, links_in, links_out, *t):
        from .ops import conv_igemm, conv_igemm_supported, pack_conv_weight_pair, dropout_seed
        xs, ws, bs = t[:n], t[n:2 * n], t[2 * n:3 * n]
        xs = [x if x.dtype == torch.bfloat16 else x.bfloat16() for x in xs]
        packs_d = [None] * n
        fb = [b.detach().float().contiguous() for b in bs]
        # Dropout behind the ReLU rides in the implicit-GEMM epilogue (mask = f(seed, branch, position), p in steps of 1/256)
        fused_drop = drop_p > 0.0 and relu and (k == 3 or _FC7_IGEMM)
        scale = 1.0
        if fused_drop:
            scale = 256.0 / (256 - min(255, int(drop_p * 256.0 + 0.5)))
        elif drop_p > 0.0:
            scale = 1.0 / (1.0 - drop_p)
        seed = dropout_seed() if fused_drop else 0
        if links_in is not None and not all(lk is not None for lk in links_in):
            links_in = None
        if k == 3:
            # both packed forms of every kernel from the float32 master in one pass each; the data-gradient form waits for backward
            need_d = any(ctx.needs_input_grad[8:8 + n]) and conv_igemm_supported(ws[0].shape[0], ws[0].shape[1], 3)
            packs = [pack_conv_weight_pair(w, True, need_d) for w in ws]
            packs_d = [p[1] for p in packs]
            outs = conv_igemm(xs, [p[0] for p in packs], fb, dils, 3, relu, drop_p if fused_drop else 0.0, seed)
        elif fused_drop:
            # fc7_k with Dropout behind it: one 1x1 implicit-GEMM launch for all branches with the mask in its epilogue beats four
            # hipBLASLt GEMMs + four dropout passes
            packs = [pack_conv_weight_pair(w, True, links_in is not None and _FUSE_CHAIN) for w in ws]
            packs_d = [p[1] for p in packs]
            outs = conv_igemm(xs, [p[0] for p in packs], fb, dils, 1, relu, drop_p, seed)
        else:
            outs = [_im2col_gemm(x, w.to(torch.bfloat16), b.to(torch.bfloat16), 1, relu) for x, w, b in zip(xs, ws, bs)]
        if drop_p > 0.0 and not fused_drop:
            outs = [torch.ops.aten.native_dropout(o, drop_p, True)[0] for o in outs]      # o = relu * mask / (1 - p)
        code, pooled = None, None
        if pool is not None:                                                             # n == 1 (conv3_3)
            from .ops import maxpool3x3_fwd
            pooled, code = maxpool3x3_fwd(outs[0], pool[0], pool[1], relu_input=True)    # the codes carry the ReLU mask
        ctx.save_for_backward(code, *xs, *ws, *(outs if (relu and pool is None) else ()))
        ctx.out_shape = tuple(outs[0].shape)
        # where a data-parallel reducer wants the weight gradients written (dsrg_amd/reducer.py: the parameters' bucket slots)
        ctx.gw_out = [getattr(w, "_dsrg_grad_out", None) for w in ws]
        ctx.packs_d = packs_d                  # not an input or output of the node: kept outside save_for_backward
        ctx.dils, ctx.relu, ctx.scale, ctx.pool, ctx.n, ctx.k = dils, relu, scale, pool, n, k
        ctx.links_in = links_in
        ctx.links_out = links_out if (relu and pool is None) else None
        if ctx.links_out is not None:
            for lk in ctx.links_out:
                lk.scale, lk.gb = scale, None
        return tuple(outs) if pool is None else (pooled,)

    @staticmethod
    def backward(ctx, *gs):
        from .ops import (conv_igemm, conv_igemm_dgrad, conv_igemm_supported, conv_igemm_wgrad, conv_igemm_wgrad_supported,
                          pack_conv_weight, relu_bwd_bias, bias_grad, maxpool3x3_bwd_relu)
        n = ctx.n
        code, saved = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        xs, ws, ys = saved[:n], saved[n:2 * n], saved[2 * n:]
        cout, cin = ws[0].shape[0], ws[0].shape[1]
        gms, gbs = [], []
        cl = torch.channels_last
        for i, g in enumerate(gs):
            lk = ctx.links_out[i] if ctx.links_out is not None else None
            gb_left = lk.take(g) if lk is not None else None
            if gb_left is not None:
                # the consumer's data gradient came masked by this node's ReLU (+ Dropout) with the bias gradient beside it
                gm, gb = g, gb_left
            elif ctx.pool is not None:
                gm, gb = maxpool3x3_bwd_relu(g, code, ctx.out_shape, ctx.pool[0])
            elif ctx.relu:
                gm, gb = relu_bwd_bias(g, ys[i], ctx.scale)
            else:
                gm = g.contiguous(memory_format=torch.channels_last)
                gb = bias_grad(gm)
            gms.append(gm); gbs.append(gb)
        need_x = [ctx.needs_input_grad[8 + i] for i in range(n)]
        gxs = [None] * n
        # inputs that are another node's ReLU outputs with no other consumer: that node's backward rides in this data gradient
        absorb = _FUSE_CHAIN and ctx.links_in is not None and all(need_x) and conv_igemm_supported(cout, cin, ctx.k) and \
            cin >= 256 and all(x.is_contiguous(memory_format=cl) for x in xs)
        if _MERGED_BWD and n == 1 and ctx.k == 3 and need_x[0] and conv_igemm_supported(cout, cin, 3) and \
                conv_igemm_wgrad_supported(cin, cout, 3) and xs[0].is_contiguous(memory_format=cl):
            # one launch for both gradients of the layer (ops.conv_igemm_backward): the data gradient's tiles and the weight gradient's
            # workgroups share a grid, so the CUs a 212-tile data gradient leaves idle do weight-gradient work
            from .ops import conv_igemm_backward
            pd = ctx.packs_d[0] if ctx.packs_d[0] is not None else pack_conv_weight(ws[0], for_dgrad=True)
            gx, gw, gb_below = conv_igemm_backward(gms[0], pd, xs[0], ctx.dils[0], xs[0] if absorb else None,
                                                   ctx.links_in[0].scale if absorb else 1.0, gw_out=ctx.gw_out[0])
            if absorb:
                ctx.links_in[0].leave(gx, gb_below)
            return (None,) * 8 + (gx, _landed(gw, ctx.gw_out[0])) + tuple(gbs)
        if absorb:
            packs_d = [p if p is not None else pack_conv_weight(w, for_dgrad=True) for p, w in zip(ctx.packs_d, ws)]
            gxs, gb_below = conv_igemm_dgrad(gms, packs_d, list(xs), ctx.dils, ctx.k, ctx.links_in[0].scale)
            for lk, gx_, gb_ in zip(ctx.links_in, gxs, gb_below):
                lk.leave(gx_, gb_)
        if ctx.k == 1:
            for i in range(n):
                if absorb:
                    break
                if need_x[i]:
                    g2d = gms[i].permute(0, 2, 3, 1).reshape(-1, cout)
                    B_, _, H_, W_ = xs[i].shape
                    gxs[i] = torch.mm(g2d, ws[i].to(torch.bfloat16).reshape(cout, cin)).view(B_, H_, W_, cin).permute(0, 3, 1, 2)
            gws = [_landed(gw, o) for gw, o in zip(conv_igemm_wgrad(list(xs), gms, ctx.dils, 1, outs=ctx.gw_out), ctx.gw_out)]
            return (None,) * 8 + tuple(gxs) + tuple(gws) + tuple(gbs)
        if any(need_x) and not absorb:
            if conv_igemm_supported(cout, cin, 3):
                packs_d = [p if p is not None else pack_conv_weight(w, for_dgrad=True) for p, w in zip(ctx.packs_d, ws)]
                gxs = conv_igemm(gms, packs_d, None, ctx.dils, 3, False)
            else:                                                                        # input channels not a multiple of 128
                for i in range(n):
                    d = ctx.dils[i]
                    gxs[i] = torch.ops.aten.convolution_backward(gms[i], xs[i], ws[i].to(torch.bfloat16), None, [1, 1], [d, d], [d, d],
                                                                 False, [0, 0], 1, [True, False, False])[0]
        if conv_igemm_wgrad_supported(cin, cout, 3):
            gws = conv_igemm_wgrad(list(xs), gms, ctx.dils, 3, outs=ctx.gw_out)          # float32, the parameters' own layout
            gws = [_landed(gw, o) for gw, o in zip(gws, ctx.gw_out)]
        else:
            gws = []
            for i in range(n):
                d = ctx.dils[i]
                gws.append(torch.ops.aten.convolution_backward(gms[i], xs[i], ws[i].to(torch.bfloat16), None, [1, 1], [d, d], [d, d],
                                                               False, [0, 0], 1, [False, True, False])[1].float())
        return (None,) * 8 + tuple(gx if nx else None for gx, nx in zip(gxs, need_x)) + tuple(gws) + tuple(gbs)


class _DirectConvFn(torch.autograd.Function):
    """conv1_1 … conv2_2 (3 / 64 / 128 channels at 321x321 and 161x161) + bias + ReLU (+ the stride-2 max pool) on the direct
    kernels with the float32 MASTER parameters as inputs, as _IgemmConvFn takes them: one pass makes both bf16 kernels (forward;
    flipped + transposed for the data gradient), the bias is read as it is, weight and bias gradients come back in float32 —
    autocast's casts (kernel and bias down, both gradients up), the flip and its copy are gone: five small launches per layer.
    apply(x, weight, bias, relu, pool, link_in, link_out); links as in _IgemmConvFn."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, pool, link_in, link_out):
        from .ops import conv3x3_direct, pack_direct_weight_pair, maxpool3x3_fwd
        x = x if x.dtype == torch.bfloat16 else x.bfloat16()
        cl = torch.channels_last
        x = x if x.is_contiguous(memory_format=cl) else x.contiguous(memory_format=cl)
        if weight.shape[1] == 3:
            w16, wd = weight.detach().to(torch.bfloat16), None                            # the image needs no gradient
        else:
            w16, wd = pack_direct_weight_pair(weight, ctx.needs_input_grad[0])
        out = conv3x3_direct(x, w16, bias.detach(), relu)
        code, pooled = None, None
        if pool is not None:
            pooled, code = maxpool3x3_fwd(out, pool[0], pool[1], relu_input=True)         # the codes carry the ReLU mask
        ctx.save_for_backward(x, out if (relu and pool is None) else None, code)
        ctx.out_shape = tuple(out.shape)
        ctx.wd = wd                                    # not an input or output of the node: kept outside save_for_backward
        ctx.gw_out = getattr(weight, "_dsrg_grad_out", None)      # a reducer's slot for the weight gradient (dsrg_amd/reducer.py)
        ctx.relu, ctx.pool = relu, pool
        ctx.link_in = link_in if (link_in is not None and link_in.scale == 1.0 and wd is not None) else None
        ctx.link_out = link_out if (relu and pool is None) else None
        if ctx.link_out is not None:
            ctx.link_out.scale, ctx.link_out.gb = 1.0, None
        return out if pool is None else pooled

    @staticmethod
    def backward(ctx, g):
        from .ops import conv3x3_direct, conv3x3_direct_dgrad, conv3x3_wgrad, relu_bwd_bias, bias_grad, maxpool3x3_bwd_relu
        x, y, code = ctx.saved_tensors
        gb = ctx.link_out.take(g) if ctx.link_out is not None else None
        if gb is not None:
            pass                                       # g came masked by this node's ReLU, its bias gradient beside it
        elif ctx.pool is not None:
            g, gb = maxpool3x3_bwd_relu(g, code, ctx.out_shape, ctx.pool[0])
        elif ctx.relu:
            g, gb = relu_bwd_bias(g, y, 1.0)
        else:
            g = g.contiguous(memory_format=torch.channels_last)
            gb = bias_grad(g)
        gx = None
        if ctx.needs_input_grad[0]:
            if _FUSE_CHAIN and ctx.link_in is not None:
                gx, gb_below = conv3x3_direct_dgrad(g, ctx.wd, x)
                ctx.link_in.leave(gx, gb_below)
            else:
                gx = conv3x3_direct(g, ctx.wd, None, False)
        gw = conv3x3_wgrad(x, g, torch.float32, out=ctx.gw_out)
        return gx, _landed(gw, ctx.gw_out), gb, None, None, None, None


def _direct_route(conv, x, p, pool):
    """does this GemmConv2d call take _DirectConvFn: one of the four narrow full-resolution layers with float32 channels_last
    master parameters under bf16 autocast, every direct kernel enabled, no Dropout, a pool only if it rides in the node"""
    from .ops import WGRAD_CONV_SHAPES
    cin, cout = conv.in_channels, conv.out_channels
    if not (_DIRECT_FN and _DIRECT_CONV == "1" and _DIRECT_C3 and _DIRECT_WGRAD and conv.kernel_size == (3, 3) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.bias is not None and (cin, cout) in WGRAD_CONV_SHAPES and p == 0.0):
        return False
    if not (x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)):
        return False
    w = conv.weight
    if not (w.dtype == torch.float32 and conv.bias.dtype == torch.float32 and w.is_contiguous(memory_format=torch.channels_last)):
        return False
    if cin == 3 and x.requires_grad and torch.is_grad_enabled():      # no direct data gradient into three channels (the image needs none)
        return False
    return pool is None or (_FUSE_POOL and 256 % (cout // 8) == 0)


_IGEMM_MIN_TILES = int(_os.environ.get("DSRG_IGEMM_MIN_TILES", "64"))


def _igemm_route(conv, x, groups=1):
    """does this GemmConv2d call take the implicit-GEMM kernels: a 3x3 'same' convolution with bias on bf16 activations
    (autocast), 64 | input channels, 256 | output channels — and enough 256 x 256 output tiles to occupy the chip: one image
    at 41x41 is 14 tiles of 72 K-steps for 256 CUs, where the library GEMM's small-M kernels win (batch-1 inference 740 against
    970 images/s); `groups` problems share the launch (the four fc6_k)"""
    if not (_IGEMM and x.is_cuda and conv.gemm and conv.kernel_size == (3, 3) and conv.bias is not None and conv.groups == 1):
        return False
    if not (x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)):
        return False
    if conv.out_channels < 256:                 # conv1_x / conv2_x: the direct kernels (weights in registers) serve the narrow layers
        return False
    pixels = x.shape[0] * x.shape[2] * x.shape[3]
    if groups * ((pixels + 255) // 256) * ((conv.out_channels + 255) // 256) < _IGEMM_MIN_TILES:
        return False
    from .ops import conv_igemm_supported
    return conv_igemm_supported(conv.in_channels, conv.out_channels, 3)


class GemmConv2d(nn.Conv2d):
    """nn.Conv2d (same parameters, same init, same state_dict) whose CUDA forward is im2col + GEMM, optionally
    with the following ReLU (`fuse_relu`) and Dropout (`fuse_dropout` = p, needs fuse_relu) fused; on the CPU it is the
    plain convolution (+ ReLU (+ Dropout)).

    Contract of the fused forms (what a user of these modules may and may not rely on):
      * `fuse_dropout` = p is realised in steps of 1/256 on the implicit-GEMM route (keep a value iff its random byte >= round(256 p);
        scale 256 / (256 - round(256 p))): p = 0.5 is exact, p = 0.3 becomes 77/256 = 0.3008.  The mask is a pure function of (seed,
        branch, position) drawn per forward from torch's generator (ops.dropout_seed), so torch.manual_seed reproduces it;
      * `chain_input=True` is the caller's PROMISE that this layer's input is the fused-ReLU output of the GemmConv2d in front of it
        and feeds nothing else.  Then the data gradient this layer returns for that activation is ALREADY masked by the lower layer's
        ReLU (and scaled by its Dropout): it is the gradient with respect to the lower layer's pre-activation, not dL/dy.  Such
        chained activations are therefore not autograd inspection points — `retain_grad()`, `torch.autograd.grad` or a tensor hook
        on them sees the masked value; a hook that RETURNS a new tensor makes the lower node fall back to its own ReLU backward
        (right on a masked gradient too) or, when a Dropout scale rides in the mask, raise (`_GradLink.take`).  Without the
        promise (the default) every activation is an ordinary autograd tensor;
      * `fuse_pool`: the node returns the pooled tensor only; the un-pooled ReLU output is not kept (the window codes carry its
        mask) and cannot be asked for."""

    def __init__(self, *args, fuse_relu=False, gemm=True, fuse_dropout=0.0, fuse_pool=None, chain_input=False, **kw):
        super().__init__(*args, **kw)
        # chain_input: the caller's promise that this layer's input is the fused-ReLU output of the GemmConv2d in front of it and
        # feeds nothing else — the data gradient may then carry that layer's ReLU backward and bias gradient (_GradLink)
        self.chain_input = bool(chain_input)
        if fuse_dropout and not fuse_relu:
            raise ValueError("fuse_dropout needs fuse_relu (the fused backward reads both masks from the output sign)")
        if fuse_pool is not None and (not fuse_relu or fuse_dropout or fuse_pool[0] != 2):
            raise ValueError("fuse_pool = (2, ceil_mode) follows a fused ReLU without dropout")
        self.fuse_relu, self.gemm, self.fuse_dropout = fuse_relu, gemm, float(fuse_dropout)
        self.fuse_pool = fuse_pool                                      # the 3x3 / stride 2 / pad 1 max pool behind the ReLU, (2, ceil_mode)

    def forward(self, x):
        p = self.fuse_dropout if self.training else 0.0
        pool = self.fuse_pool
        if x.is_cuda and self.stride == (1, 1) and self.kernel_size[0] in (1, 3) and \
                self.padding[0] == self.dilation[0] * (self.kernel_size[0] // 2):
            # the pool rides inside the node when its kernels apply: bf16 activations (autocast), 8 | channels, (channels / 8) | 256
            cout = self.out_channels
            if _igemm_route(self, x):
                in_node = pool is not None and _FUSE_POOL and cout % 8 == 0 and 256 % (cout // 8) == 0     # the pool kernels' tiling
                lin = getattr(x, "_dsrg_grad_link", None) if self.chain_input else None
                lout = _GradLink() if (self.fuse_relu and pool is None and torch.is_grad_enabled()) else None
                (out,) = _IgemmConvFn.apply(3, [self.dilation[0]], self.fuse_relu, p, pool if in_node else None, 1,
                                            [lin] if lin is not None else None, [lout] if lout is not None else None, x, self.weight,
                                            self.bias)
                if lout is not None:
                    out._dsrg_grad_link = lout
                return out if pool is None or in_node else _pool3x3(out, pool[0], pool[1])
            if _direct_route(self, x, p, pool):
                lin = getattr(x, "_dsrg_grad_link", None) if self.chain_input else None
                lout = _GradLink() if (self.fuse_relu and pool is None and torch.is_grad_enabled()) else None
                out = _DirectConvFn.apply(x, self.weight, self.bias, self.fuse_relu, pool, lin, lout)
                if lout is not None:
                    out._dsrg_grad_link = lout
                return out
            in_node = pool is not None and _FUSE_POOL and cout % 8 == 0 and 256 % (cout // 8) == 0 and self.bias is not None and (
                x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16))
            lin = getattr(x, "_dsrg_grad_link", None) if self.chain_input else None
            lout = _GradLink() if (self.fuse_relu and pool is None and p == 0.0 and torch.is_grad_enabled()) else None
            out = _ConvFn.apply(x, self.weight, self.bias, self.dilation[0], self.fuse_relu, self.gemm, p, pool if in_node else None,
                                lin, lout)
            if lout is not None:
                out._dsrg_grad_link = lout
            return out if pool is None or in_node else _pool3x3(out, pool[0], pool[1])
        out = super().forward(x)
        out = F.relu(out) if self.fuse_relu else out
        out = F.dropout(out, p, True) if p > 0.0 else out
        return out if pool is None else _pool3x3(out, pool[0], pool[1])


class FusedReLU(nn.Identity):
    """placeholder that keeps the Sequential indices (and state_dict keys) of the conv/ReLU pairs: the ReLU itself
    runs inside the GemmConv2d in front of it"""


class FusedPool(nn.Identity):
    """placeholder for the max-pool layer that runs inside the GemmConv2d two slots in front of it (`fuse_pool`)"""


class FusedDropout(nn.Identity):
    """placeholder for the Dropout layer that runs inside the GemmConv2d two slots in front of it (`fuse_dropout`)"""


class _MaxPool3x3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stride, ceil_mode):
        from .ops import maxpool3x3_fwd
        out, code = maxpool3x3_fwd(x, stride, ceil_mode)
        ctx.save_for_backward(code)
        ctx.in_shape, ctx.stride = x.shape, stride
        return out

    @staticmethod
    def backward(ctx, g):
        from .ops import maxpool3x3_bwd
        (code,) = ctx.saved_tensors
        return maxpool3x3_bwd(g, code, ctx.in_shape, ctx.stride), None, None


class _AvgPool3x3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        from .ops import avgpool3x3_s1
        return avgpool3x3_s1(x)

    @staticmethod
    def backward(ctx, g):
        from .ops import avgpool3x3_s1
        return avgpool3x3_s1(g)                                          # symmetric stencil: its own adjoint


class AvgPool3x3(nn.AvgPool2d):
    """pool5a: 3x3 / stride 1 / pad 1 average over padded windows (Caffe AVE); one HIP pass each way for bf16
    channels_last activations, nn.AvgPool2d otherwise.  Not only a speed matter: PyTorch-ROCm 2.10's avg_pool2d
    backward returns wrong gradients for channels_last inputs (tests/test_gpu_parity.py checks ours against the fp32
    NCHW adjoint), so any other channels_last input is made contiguous first."""

    def __init__(self):
        super().__init__(3, 1, 1)

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.bfloat16 and x.shape[1] % 8 == 0 and \
                x.is_contiguous(memory_format=torch.channels_last):
            return _AvgPool3x3Fn.apply(x)
        if x.is_cuda and x.dim() == 4 and not x.is_contiguous():
            x = x.contiguous()                                            # keep torch's kernel on its correct (NCHW) path
        return super().forward(x)


class MaxPool3x3(nn.MaxPool2d):
    """3x3 / pad 1 max pooling (stride 1 or 2): one HIP pass each way over bf16 channels_last activations with
    1-byte window codes instead of torch's int64 indices; anything else takes nn.MaxPool2d's path."""

    def __init__(self, stride, ceil_mode=False):
        super().__init__(3, stride, 1, ceil_mode=ceil_mode)

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.bfloat16 and x.shape[1] % 8 == 0 and \
                x.is_contiguous(memory_format=torch.channels_last):
            return _MaxPool3x3Fn.apply(x, self.stride, self.ceil_mode)
        return super().forward(x)


def _pool3x3(x, stride, ceil_mode):
    """3x3 / pad 1 max pool outside a conv node: the HIP pass for bf16 channels_last activations, torch's otherwise"""
    if x.is_cuda and x.dtype == torch.bfloat16 and x.shape[1] % 8 == 0 and x.is_contiguous(memory_format=torch.channels_last):
        return _MaxPool3x3Fn.apply(x, stride, ceil_mode)
    return F.max_pool2d(x, 3, stride, 1, ceil_mode=ceil_mode)


def _conv_relu(cin, cout, dilation=1, gemm=False, pool=None, chain=False):
    """conv + ReLU (+ the stride-2 max pool behind them, `pool` = (2, ceil_mode)): Sequential slots as in the prototxt;
    chain: the input is the conv + ReLU in front and nothing else reads it (GemmConv2d.chain_input)"""
    conv = GemmConv2d(cin, cout, 3, padding=dilation, dilation=dilation, fuse_relu=True, gemm=gemm, fuse_pool=pool, chain_input=chain)
    return [conv, FusedReLU()] + ([FusedPool()] if pool is not None else [])


class _HeadsFn(torch.autograd.Function):
    """The four fc8-SEC_k classifiers + Eltwise SUM on bf16 activations with fp32 weights, fp32 accumulation and an fp32
    NCHW result (one HIP pass on the f32-input MFMA, ops.heads_forward).  Backward (ops.heads_backward): the gradient that
    continues into the bf16 trunk is accumulated in fp32 and rounded once; weight and bias gradients are fp32."""

    @staticmethod
    def forward(ctx, links, weight, bias, *xs):
        from .ops import heads_forward
        cl = torch.channels_last
        # links: the x_k are ReLU (+ Dropout) outputs of nodes that feed these classifiers only (their backward rides along)
        ctx.links = links if (links is not None and all(lk is not None for lk in links) and
                              all(x.is_contiguous(memory_format=cl) for x in xs)) else None
        xs = [x.contiguous(memory_format=cl) for x in xs]
        weight = weight.contiguous()
        ctx.save_for_backward(weight, *xs)
        return heads_forward(xs, weight, bias.contiguous())

    @staticmethod
    def backward(ctx, g):
        from .ops import heads_backward
        weight, *xs = ctx.saved_tensors
        n, O, K = weight.shape
        if _FUSE_CHAIN and ctx.links is not None and all(ctx.needs_input_grad[3:]):
            gxs, gw, gb_below = heads_backward(xs, weight, g, True, ctx.links[0].scale)
            for i, lk in enumerate(ctx.links):
                lk.leave(gxs[i], gb_below[i])
        else:
            gxs, gw = heads_backward(xs, weight, g, need_gx=any(ctx.needs_input_grad[3:]))
        gb1 = g.sum((0, 2, 3))                                            # fp32, the same for every branch
        return (None, gw, gb1.unsqueeze(0).expand(n, O).contiguous()) + (tuple(gxs) if gxs is not None else (None,) * n)


class _StackKernels(torch.autograd.Function):
    """the (cout, cin, 1, 1) classifier kernels as one (n, cout, cin) tensor; backward hands every kernel its slice of the gradient
    in the PARAMETER's strides (a channels_last 1x1 kernel has strides (cin, 1, cin, cin); the plain reshape's gradient has
    (cin, 1, 1, 1) — the same memory, but DistributedDataParallel compares strides, warns and copies the gradient into its bucket)"""

    @staticmethod
    def forward(ctx, *ws):
        ctx.layouts = [(tuple(w.shape), tuple(w.stride())) for w in ws]
        return torch.stack([w.reshape(w.shape[0], w.shape[1]) for w in ws])

    @staticmethod
    def backward(ctx, g):
        return tuple(g[i].as_strided(shape, stride) if g[i].is_contiguous() else g[i].reshape(shape)
                     for i, (shape, stride) in enumerate(ctx.layouts))


class VGG16ASPP(nn.Module):
    """features (conv1_1 .. conv5_3, pools, pool5a) + four ASPP branches fc6_k -> fc7_k -> fc8-SEC_k, summed.  On a GPU under
    bf16 autocast conv1_2, conv2_2, conv3_2 .. conv5_3, fc7_k and the classifiers are built with `chain_input` (see GemmConv2d):
    the activations between them are not autograd inspection points, and Dropout's p is rounded to a multiple of 1/256 (the
    prototxt's 0.5 is exact).  `DSRG_FUSE_CHAIN=0` restores one ordinary autograd node per layer."""

    def __init__(self, num_classes=21, dropout=0.5, gemm_convs=True):
        super().__init__()
        L = []
        p2 = (2, True)                                                  # pool1-3: 3x3 / stride 2 / pad 1, ceil mode (Caffe), inside the conv node
        L += _conv_relu(3, 64) + _conv_relu(64, 64, pool=p2, chain=True)
        L += _conv_relu(64, 128, 1, gemm_convs) + _conv_relu(128, 128, 1, gemm_convs, pool=p2, chain=True)
        L += _conv_relu(128, 256, 1, gemm_convs) + _conv_relu(256, 256, 1, gemm_convs, chain=True) + \
            _conv_relu(256, 256, 1, gemm_convs, pool=p2, chain=True)
        g = gemm_convs                                                  # the 41x41 stages
        L += _conv_relu(256, 512, 1, g) + _conv_relu(512, 512, 1, g, chain=True) + _conv_relu(512, 512, 1, g, chain=True) + [MaxPool3x3(1)]
        L += _conv_relu(512, 512, 2, g) + _conv_relu(512, 512, 2, g, chain=True) + _conv_relu(512, 512, 2, g, chain=True) + [MaxPool3x3(1)]
        L += [AvgPool3x3()]                                   # pool5a AVE (count_include_pad, as Caffe)
        self.features = nn.Sequential(*L)
        self.branches = nn.ModuleList()
        for d in (6, 12, 18, 24):
            g = gemm_convs
            fc8 = GemmConv2d(1024, num_classes, 1, gemm=g)
            nn.init.normal_(fc8.weight, std=0.01)
            nn.init.zeros_(fc8.bias)
            self.branches.append(nn.Sequential(
                GemmConv2d(512, 1024, 3, padding=d, dilation=d, fuse_relu=True, gemm=g, fuse_dropout=dropout), FusedReLU(),
                FusedDropout(),
                GemmConv2d(1024, 1024, 1, fuse_relu=True, gemm=g, fuse_dropout=dropout), FusedReLU(), FusedDropout(), fc8))

    def forward(self, x):
        """-> fc8-SEC scores (B,21,h,w) in float32.  Under autocast the trunk and fc6/fc7 run in the autocast dtype, but
        the four fc8-SEC_k classifiers and their Eltwise SUM (train-s.prototxt:737-744) always run in float32 (fp32
        weights, fp32 accumulation, fp32 sum): these scores feed hard decisions the reference takes on fp32/fp64 values
        (Softmax + 1e-4, the CRF, the 0.85 / 0.99 SRG thresholds, the 0.05 / 20 clip of the constrain loss), and a bf16
        score of magnitude 8-16 has a step of 0.06-0.125."""
        f = self.features(x)
        hs = []
        fc6 = [br[0] for br in self.branches]
        if len(fc6) <= 4 and all(_igemm_route(m, f, len(fc6)) and m.stride == (1, 1) and m.padding == m.dilation for m in fc6):
            # the four fc6_k (same input, own dilation) in one launch each way: 1696 tiles fill the chip where 424 leave a sixth idle
            p = fc6[0].fuse_dropout if self.training else 0.0
            n = len(fc6)
            links = [_GradLink() for _ in range(n)] if torch.is_grad_enabled() else None
            hs = list(_IgemmConvFn.apply(3, [m.dilation[0] for m in fc6], True, p, None, n, None, links, *([f] * n),
                                         *[m.weight for m in fc6], *[m.bias for m in fc6]))
            fc7 = [br[3] for br in self.branches]
            if all(isinstance(m, GemmConv2d) and m.kernel_size == (1, 1) and m.fuse_relu and m.bias is not None and m.gemm
                   and m.in_channels % 256 == 0 and m.out_channels % 256 == 0 for m in fc7):
                # fc7_k: own input each, one launch for the four weight gradients
                p7 = fc7[0].fuse_dropout if self.training else 0.0
                # (fc6_k's output feeds fc7_k only: its ReLU / Dropout backward and bias gradient ride in fc7_k's data gradient)
                links7 = [_GradLink() for _ in range(n)] if torch.is_grad_enabled() else None
                hs = list(_IgemmConvFn.apply(1, [1] * n, True, p7, None, n, links, links7, *hs, *[m.weight for m in fc7],
                                             *[m.bias for m in fc7]))
                for h, lk in zip(hs, links7 or []):
                    h._dsrg_grad_link = lk
            else:
                for i, br in enumerate(self.branches):
                    for m in list(br)[1:-1]:
                        hs[i] = m(hs[i])
        else:
            for br in self.branches:
                h = f
                for m in list(br)[:-1]:
                    h = m(h)
                hs.append(h)
        heads = [br[-1] for br in self.branches]
        if hs[0].is_cuda and hs[0].dtype == torch.bfloat16 and len(hs) <= 4 and heads[0].out_channels <= 32 \
                and heads[0].in_channels % 256 == 0:
            w = _StackKernels.apply(*[m.weight for m in heads])
            b = torch.stack([m.bias for m in heads])
            # (on the grouped route the fc7_k outputs feed these classifiers and nothing else)
            links = [getattr(h, "_dsrg_grad_link", None) for h in hs]
            return _HeadsFn.apply(links, w.float(), b.float(), *hs)
        out = None
        with torch.autocast(device_type=hs[0].device.type, enabled=False):
            for m, h in zip(heads, hs):
                s = m(h.float())
                out = s if out is None else out + s
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


class GraphedForward(object):
    """An eval-mode forward of `net` captured once into a hipGraph (torch.cuda.CUDAGraph) and replayed per call.  At batch 1
    the VGG16-ASPP forward is ~150 launches of 3-30 us and launch-bound (1.6 ms eager, 0.7 ms replayed); its shapes, weights
    and buffers are static at test time, and every HIP kernel of this package launches on torch's current stream, which is
    the capture stream.  The eager warm-up runs first so that TunableOp has picked its GEMM solutions and the kernels have
    reserved their LDS outside the capture.  __call__(x) copies x into the captured input buffer and returns the captured
    output buffer (overwritten by the next call)."""

    def __init__(self, net, example, amp_dtype=torch.bfloat16, warmup=3):
        if net.training:
            raise ValueError("GraphedForward captures an eval-mode forward (dropout would replay one mask)")
        self.net, self.amp = net, amp_dtype
        self.x = example.detach().clone(memory_format=torch.preserve_format)
        warm = torch.cuda.Stream(device=example.device)
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):
            for _ in range(warmup):
                self._fwd()
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._fwd()

    def _fwd(self):
        with torch.no_grad(), torch.autocast("cuda", dtype=self.amp, enabled=self.amp is not None):
            return self.net(self.x).contiguous()

    def __call__(self, x=None):
        if x is not None and x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x)
        self.graph.replay()
        return self.out
