// Backbone plumbing around the hipBLASLt / MIOpen convolutions of the VGG16-ASPP net (no reference
// counterpart: the reference's ReLU / Pooling layers live in the external Caffe framework, train-s.prototxt:41-744).
// All three are HBM-bound elementwise passes over NHWC bf16 activations, 16 bytes (8 channels) per lane:
//   relu_bwd_bias : ReLU backward fused with the per-channel bias-gradient reduction of the preceding convolution
//   maxpool3x3    : 3x3 max pooling (stride 1 or 2, pad 1, optional ceil mode) forward with a 1-byte window code,
//                   and the gather-form backward that reads the codes (no atomics, deterministic)
#include "common.h"
#include <algorithm>
#ifndef DSRG_EXP
#define DSRG_EXP 0                // experiment builds (Makefile EXP= EXPSRC=backbone_ops; tools only)
#endif
#include <cstdlib>
#include <cstring>

namespace dsrg {

__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;      // NaN stays NaN
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) { return f32_to_bf16_rne(lo) | (f32_to_bf16_rne(hi) << 16); }

constexpr int kRbThreads = 256;

// gm = scale * g * (y > 0); part[blockIdx][c] = sum over this block's rows of gm[:, c]   (rows x C, C % 8 == 0, C/8 <= 256)
// scale = 1 for a plain ReLU; 1/(1-p) when y is the output of ReLU followed by dropout (then y > 0 is both masks at once)
__global__ __launch_bounds__(kRbThreads) void relu_bwd_bias_kernel(const uint4 *__restrict__ g, const uint4 *__restrict__ y,
                                                                    uint4 *__restrict__ gm, float *__restrict__ part,
                                                                    int rows, int C8, int rows_per_block, float scale) {
    __shared__ float red[kRbThreads][9];                                 // +1 pad: column reads hit distinct banks
    const int lanes_r = kRbThreads / C8;                                 // row lanes per block
    const int cg = threadIdx.x % C8, rl = threadIdx.x / C8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (rl < lanes_r) {
        const int r0 = blockIdx.x * rows_per_block;
        const int r1 = min(rows, r0 + rows_per_block);
        for (int r = r0 + rl; r < r1; r += lanes_r) {
            const size_t i = (size_t)r * C8 + cg;
            const uint4 gv = g[i];
            // y == nullptr: no ReLU in front (plain column sums of g, nothing stored); 0x3f80 = bf16 1.0 passes the mask
            const uint4 yv = y ? y[i] : make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
            const uint32_t gw[4] = {gv.x, gv.y, gv.z, gv.w}, yw[4] = {yv.x, yv.y, yv.z, yv.w};
            uint32_t ow[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // y is a ReLU output: positive <=> non-zero magnitude with a clear sign bit (NaN passes the gradient, as torch)
                const bool plo = (yw[k] & 0x8000u) == 0 && (yw[k] & 0x7fffu) != 0;
                const bool phi = (yw[k] & 0x80000000u) == 0 && (yw[k] & 0x7fff0000u) != 0;
                const uint32_t m = (plo ? 0xffffu : 0u) | (phi ? 0xffff0000u : 0u);
                ow[k] = gw[k] & m;
                if (scale != 1.0f) ow[k] = pack_bf16(bf16_lo(ow[k]) * scale, bf16_hi(ow[k]) * scale);
                acc[2 * k] += bf16_lo(ow[k]);
                acc[2 * k + 1] += bf16_hi(ow[k]);
            }
            if (gm) gm[i] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[threadIdx.x][k] = acc[k];
    __syncthreads();
    // thread t < C sums channel t over the row lanes in a fixed order
    for (int c = threadIdx.x; c < C8 * 8; c += kRbThreads) {
        const int g8 = c >> 3, k = c & 7;
        float s = 0.f;
        for (int l = 0; l < lanes_r; ++l) s += red[l * C8 + g8][k];
        part[(size_t)blockIdx.x * (C8 * 8) + c] = s;
    }
}

// 32 channels x 8 slices of the partial rows per block; slice sums are combined in slice order (fixed order => deterministic)
__global__ __launch_bounds__(256) void bias_finalize_kernel(const float *__restrict__ part, float *__restrict__ bias_grad,
                                                            int nblk, int C) {
    __shared__ float red[8][33];
    bias_finalize_block(part, bias_grad, nblk, C, (int)blockIdx.x, red);
}

int launch_relu_bwd_bias(const void *g, const void *y, void *gm, float *bias_grad, float *part, int part_blocks,
                         long rows, int C, float scale, hipStream_t stream) {
    if (C % 8 != 0 || C / 8 > kRbThreads || C < 8) return set_error(DSRG_ERR_UNSUPPORTED, "relu_bwd_bias: channels must be a multiple of 8, at most 2048");
    if (rows <= 0 || rows > 0x7fffffffL) return set_error(DSRG_ERR_INVALID, "relu_bwd_bias: bad row count");
    if (part_blocks < 1) return set_error(DSRG_ERR_INVALID, "relu_bwd_bias: no partial-sum blocks");
    const int C8 = C / 8;
    int nblk = part_blocks;
    int rpb = (int)((rows + nblk - 1) / nblk);
    const int lanes_r = kRbThreads / C8;
    rpb = ((rpb + lanes_r - 1) / lanes_r) * lanes_r;
    nblk = (int)((rows + rpb - 1) / rpb);
    hipLaunchKernelGGL(relu_bwd_bias_kernel, dim3(nblk), dim3(kRbThreads), 0, stream, (const uint4 *)g, (const uint4 *)y,
                       (uint4 *)gm, part, (int)rows, C8, rpb, scale);
    DSRG_LAUNCH_CHECK();
    if (!defer_reduction(1, part, bias_grad, nblk, C))
        hipLaunchKernelGGL(bias_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, stream, part, bias_grad, nblk, C);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

// ---- 3x3 max pooling, NHWC bf16 ----------------------------------------------------------------------------------
// forward: out[b,oy,ox,c] = max over the window clipped to the image; code = 3*dy+dx of the FIRST maximum in row-major
// window order (the rule of Caffe's PoolingLayer and of torch's max_pool2d); NaN propagates as in torch.
// RELU_IN: the input is a ReLU's output and the pool's backward is to carry that ReLU's backward too: a window whose maximum is
// not positive gets the code kPoolDead, which matches no tap — its gradient goes nowhere, exactly what masking the pooled-back
// gradient with (input > 0) does afterwards (a pixel receives gradient only as the argmax of a window, and its value IS that
// window's maximum), so the backward never has to read the pool's input again (211 MB at pool1).
constexpr uint32_t kPoolDead = 0xfeu;
template <bool RELU_IN>
__global__ void maxpool3x3_fwd_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, uint2 *__restrict__ code,
                                      int B, int H, int W, int OH, int OW, int C8, int stride) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * OH * OW * C8;
    if (idx >= total) return;
    const int c = (int)(idx % C8);
    size_t r = idx / C8;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const int b = (int)(r / OH);
    float best[8];
    uint32_t bc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { best[k] = -__builtin_inff(); bc[k] = 0xffu; }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = oy * stride - 1 + dy;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int xx = ox * stride - 1 + dx;
            if (xx < 0 || xx >= W) continue;
            const uint4 v = in[(((size_t)b * H + yy) * W + xx) * C8 + c];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float lo = bf16_lo(w[k]), hi = bf16_hi(w[k]);
                if (lo > best[2 * k] || lo != lo || bc[2 * k] == 0xffu) { best[2 * k] = lo; bc[2 * k] = 3 * dy + dx; }
                if (hi > best[2 * k + 1] || hi != hi || bc[2 * k + 1] == 0xffu) { best[2 * k + 1] = hi; bc[2 * k + 1] = 3 * dy + dx; }
            }
        }
    }
    out[idx] = make_uint4(pack_bf16(best[0], best[1]), pack_bf16(best[2], best[3]), pack_bf16(best[4], best[5]),
                          pack_bf16(best[6], best[7]));
    if (RELU_IN) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (best[k] <= 0.0f) bc[k] = kPoolDead;                          // (+0, -0, negatives; a NaN keeps its window)
    }
    code[idx] = make_uint2(bc[0] | (bc[1] << 8) | (bc[2] << 16) | (bc[3] << 24), bc[4] | (bc[5] << 8) | (bc[6] << 16) | (bc[7] << 24));
}

// backward, gather form: every input pixel visits the <= 9 (stride 1) or <= 4 (stride 2) windows that contain it and
// takes the window's gradient where the window's code names this pixel.  Sums in f32 in a fixed order.
__global__ void maxpool3x3_bwd_kernel(const uint4 *__restrict__ gout, const uint2 *__restrict__ code, uint4 *__restrict__ gin,
                                      int B, int H, int W, int OH, int OW, int C8, int stride) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * H * W * C8;
    if (idx >= total) return;
    const int c = (int)(idx % C8);
    size_t r = idx / C8;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    const int b = (int)(r / H);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int ty = y + 1 - dy;                                        // oy*stride - 1 + dy == y
        if (ty < 0 || ty % stride != 0) continue;
        const int oy = ty / stride;
        if (oy >= OH) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int tx = x + 1 - dx;
            if (tx < 0 || tx % stride != 0) continue;
            const int ox = tx / stride;
            if (ox >= OW) continue;
            const size_t o = (((size_t)b * OH + oy) * OW + ox) * C8 + c;
            const uint2 cd = code[o];
            const uint4 gv = gout[o];
            const uint32_t want = 3 * dy + dx;
            const uint32_t gw[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t cw = k < 2 ? cd.x : cd.y;
                const uint32_t c0 = (cw >> (16 * (k & 1))) & 0xffu, c1 = (cw >> (16 * (k & 1) + 8)) & 0xffu;
                if (c0 == want) acc[2 * k] += bf16_lo(gw[k]);
                if (c1 == want) acc[2 * k + 1] += bf16_hi(gw[k]);
            }
        }
    }
    gin[idx] = make_uint4(pack_bf16(acc[0], acc[1]), pack_bf16(acc[2], acc[3]), pack_bf16(acc[4], acc[5]), pack_bf16(acc[6], acc[7]));
}

// backward for stride 2, one thread per 2 x 2 block of input pixels (x 8 channels): the block (2 y2 .. + 1, 2 x2 .. + 1) meets
// exactly the four windows (y2 .. y2 + 1, x2 .. x2 + 1), so their gradients and codes are loaded once per block instead of
// 2.25 times per pixel (the gather kernel above is bound by those loads: 162 us for pool1 at batch 16).  Same sums in the
// same order as the gather form (dy ascending, then dx).
__global__ void maxpool3x3_s2_bwd_kernel(const uint4 *__restrict__ gout, const uint2 *__restrict__ code, uint4 *__restrict__ gin,
                                         int B, int H, int W, int OH, int OW, int C8) {
    const int H2 = (H + 1) >> 1, W2 = (W + 1) >> 1;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * H2 * W2 * C8;
    if (idx >= total) return;
    const int c = (int)(idx % C8);
    size_t r = idx / C8;
    const int x2 = (int)(r % W2);
    r /= W2;
    const int y2 = (int)(r % H2);
    const int b = (int)(r / H2);
    uint32_t gw[2][2][4], cw[2][2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            uint4 gv = make_uint4(0u, 0u, 0u, 0u);
            uint2 cd = make_uint2(0xffffffffu, 0xffffffffu);                 // no window: matches no tap
            if (y2 + i < OH && x2 + j < OW) {
                const size_t o = (((size_t)b * OH + y2 + i) * OW + x2 + j) * C8 + c;
                cd = code[o];
                gv = gout[o];
            }
            gw[i][j][0] = gv.x; gw[i][j][1] = gv.y; gw[i][j][2] = gv.z; gw[i][j][3] = gv.w;
            cw[i][j][0] = cd.x; cw[i][j][1] = cd.y;
        }
    auto take = [&](float (&acc)[8], int i, int j, uint32_t want) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t w = cw[i][j][k >> 1];
            const uint32_t c0 = (w >> (16 * (k & 1))) & 0xffu, c1 = (w >> (16 * (k & 1) + 8)) & 0xffu;
            if (c0 == want) acc[2 * k] += bf16_lo(gw[i][j][k]);
            if (c1 == want) acc[2 * k + 1] += bf16_hi(gw[i][j][k]);
        }
    };
    auto store = [&](const float (&acc)[8], int y, int x) {
        if (y < H && x < W)
            gin[(((size_t)b * H + y) * W + x) * C8 + c] =
                make_uint4(pack_bf16(acc[0], acc[1]), pack_bf16(acc[2], acc[3]), pack_bf16(acc[4], acc[5]), pack_bf16(acc[6], acc[7]));
    };
    const int y0 = 2 * y2, x0 = 2 * x2;
    {   // (even, even): the centre of window (y2, x2)
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        take(acc, 0, 0, 4u);
        store(acc, y0, x0);
    }
    {   // (even, odd): left column of window (y2, x2 + 1), right column of (y2, x2)
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        take(acc, 0, 1, 3u);
        take(acc, 0, 0, 5u);
        store(acc, y0, x0 + 1);
    }
    {   // (odd, even): top row of window (y2 + 1, x2), bottom row of (y2, x2)
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        take(acc, 1, 0, 1u);
        take(acc, 0, 0, 7u);
        store(acc, y0 + 1, x0);
    }
    {   // (odd, odd): a corner of all four
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        take(acc, 1, 1, 0u);
        take(acc, 1, 0, 2u);
        take(acc, 0, 1, 6u);
        take(acc, 0, 0, 8u);
        store(acc, y0 + 1, x0 + 1);
    }
}

// The same with the ReLU backward of the layer in front of the pool and its bias gradient fused in (conv + ReLU + pool is
// how conv1_2, conv2_2 and conv3_3 sit in train-s.prototxt:65-226): gin = (y > 0) ? pooled-back gradient : 0, written once, and
// part[block][c] = this block's column sums of gin — instead of writing the unmasked gradient and passing over it again with
// relu_bwd_bias_kernel (three more passes over the largest activations of the net).  A block owns a contiguous range of
// (2 x 2 pixel block, channel group) items, 256 per round, so a thread keeps its channel group (256 % C8 == 0).
// HAS_Y = false: the codes come from maxpool3x3_fwd_kernel<true> and carry the mask already (y is not read: same bits).
template <bool HAS_Y>
__global__ __launch_bounds__(kRbThreads) void maxpool3x3_s2_bwd_relu_kernel(const uint4 *__restrict__ gout, const uint2 *__restrict__ code,
                                                                             const uint4 *__restrict__ y, uint4 *__restrict__ gin,
                                                                             float *__restrict__ part, int B, int H, int W, int OH,
                                                                             int OW, int C8, size_t items_per_block) {
    __shared__ float red[kRbThreads][9];
    const int H2 = (H + 1) >> 1, W2 = (W + 1) >> 1;
    const size_t total = (size_t)B * H2 * W2 * C8;
    const size_t beg = (size_t)blockIdx.x * items_per_block, end = beg + items_per_block < total ? beg + items_per_block : total;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (size_t idx = beg + threadIdx.x; idx < end; idx += kRbThreads) {
        const int c = (int)(idx % C8);
        size_t r = idx / C8;
        const int x2 = (int)(r % W2);
        r /= W2;
        const int y2 = (int)(r % H2);
        const int b = (int)(r / H2);
        uint32_t gw[2][2][4], cw[2][2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint4 gv = make_uint4(0u, 0u, 0u, 0u);
                uint2 cd = make_uint2(0xffffffffu, 0xffffffffu);             // no window: matches no tap
                if (y2 + i < OH && x2 + j < OW) {
                    const size_t o = (((size_t)b * OH + y2 + i) * OW + x2 + j) * C8 + c;
                    cd = code[o];
                    gv = gout[o];
                }
                gw[i][j][0] = gv.x; gw[i][j][1] = gv.y; gw[i][j][2] = gv.z; gw[i][j][3] = gv.w;
                cw[i][j][0] = cd.x; cw[i][j][1] = cd.y;
            }
        auto take = [&](float (&acc)[8], int i, int j, uint32_t want) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t w = cw[i][j][k >> 1];
                const uint32_t c0 = (w >> (16 * (k & 1))) & 0xffu, c1 = (w >> (16 * (k & 1) + 8)) & 0xffu;
                if (c0 == want) acc[2 * k] += bf16_lo(gw[i][j][k]);
                if (c1 == want) acc[2 * k + 1] += bf16_hi(gw[i][j][k]);
            }
        };
        auto store = [&](const float (&acc)[8], int yy, int xx) {
            if (yy >= H || xx >= W) return;
            const size_t i = (((size_t)b * H + yy) * W + xx) * C8 + c;
            uint4 yv = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
            if (HAS_Y) yv = y[i];
            const uint32_t yw[4] = {yv.x, yv.y, yv.z, yv.w};
            uint32_t ow[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool plo = (yw[k] & 0x8000u) == 0 && (yw[k] & 0x7fffu) != 0;          // as relu_bwd_bias_kernel
                const bool phi = (yw[k] & 0x80000000u) == 0 && (yw[k] & 0x7fff0000u) != 0;
                const uint32_t m = (plo ? 0xffffu : 0u) | (phi ? 0xffff0000u : 0u);
                ow[k] = pack_bf16(acc[2 * k], acc[2 * k + 1]) & m;
                bsum[2 * k] += bf16_lo(ow[k]);
                bsum[2 * k + 1] += bf16_hi(ow[k]);
            }
            gin[i] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        };
        const int y0 = 2 * y2, x0 = 2 * x2;
        {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            take(acc, 0, 0, 4u);
            store(acc, y0, x0);
        }
        {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            take(acc, 0, 1, 3u);
            take(acc, 0, 0, 5u);
            store(acc, y0, x0 + 1);
        }
        {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            take(acc, 1, 0, 1u);
            take(acc, 0, 0, 7u);
            store(acc, y0 + 1, x0);
        }
        {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            take(acc, 1, 1, 0u);
            take(acc, 1, 0, 2u);
            take(acc, 0, 1, 6u);
            take(acc, 0, 0, 8u);
            store(acc, y0 + 1, x0 + 1);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[threadIdx.x][k] = bsum[k];
    __syncthreads();
    const int lanes = kRbThreads / C8;                                       // threads that share a channel group: t % C8 equal
    for (int ch = threadIdx.x; ch < C8 * 8; ch += kRbThreads) {
        const int g8 = ch >> 3, k = ch & 7;
        float s2 = 0.f;
        for (int l = 0; l < lanes; ++l) s2 += red[l * C8 + g8][k];
        part[(size_t)blockIdx.x * (C8 * 8) + ch] = s2;
    }
}

// ---- 3x3 / stride 1 / pad 1 average pooling over padded windows (Caffe AVE pooling = count_include_pad), NHWC bf16 --------
// out = (sum of the in-image taps) / 9.  The stencil is symmetric, so the backward pass is the same kernel on the gradient.
__global__ void avgpool3x3_s1_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, int B, int H, int W, int C8) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * H * W * C8;
    if (idx >= total) return;
    const int c = (int)(idx % C8);
    size_t r = idx / C8;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    const int b = (int)(r / H);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= W) continue;
            const uint4 v = in[(((size_t)b * H + yy) * W + xx) * C8 + c];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc[2 * k] += bf16_lo(w[k]); acc[2 * k + 1] += bf16_hi(w[k]); }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = acc[k] / 9.0f;
    out[idx] = make_uint4(pack_bf16(acc[0], acc[1]), pack_bf16(acc[2], acc[3]), pack_bf16(acc[4], acc[5]), pack_bf16(acc[6], acc[7]));
}
int launch_avgpool3x3_s1(const void *in, void *out, int B, int H, int W, int C, hipStream_t stream) {
    if (C % 8 != 0) return set_error(DSRG_ERR_UNSUPPORTED, "avgpool3x3: channels must be a multiple of 8");
    if (B <= 0 || H <= 0 || W <= 0) return set_error(DSRG_ERR_INVALID, "avgpool3x3: bad shape");
    const size_t total = (size_t)B * H * W * (C / 8);
    hipLaunchKernelGGL(avgpool3x3_s1_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const uint4 *)in,
                       (uint4 *)out, B, H, W, C / 8);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

// ---- the tail of a ResNet bottleneck (train-f stage on the DeepLab-v2 ResNet-101, BASELINE.json configs[4]; no reference
// counterpart): y = relu(a + b) in one pass over bf16 tensors (fp32 sum, one rounding) instead of an add and a threshold pass, and
// its backward gm = (y > 0 ? g (+ g2) : 0) — g2: a second gradient of y that autograd would otherwise add in a pass of its own
// (the block's output feeds the next block's first convolution AND its identity path).  Flat arrays, 16 bytes per thread.
__global__ __launch_bounds__(256) void add_relu_kernel(const uint4 *__restrict__ a, const uint4 *__restrict__ b, uint4 *__restrict__ y, size_t n8) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const uint4 va = a[i], vb = b[i];
    const uint32_t wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
        o[k] = pack_bf16(fmaxf(bf16_lo(wa[k]) + bf16_lo(wb[k]), 0.0f), fmaxf(bf16_hi(wa[k]) + bf16_hi(wb[k]), 0.0f));
    y[i] = make_uint4(o[0], o[1], o[2], o[3]);
}
__global__ __launch_bounds__(256) void relu_mask_kernel(const uint4 *__restrict__ g, const uint4 *__restrict__ g2, const uint4 *__restrict__ y,
                                                        uint4 *__restrict__ gm, size_t n8) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const uint4 vg = g[i], vy = y[i];
    uint32_t wg[4] = {vg.x, vg.y, vg.z, vg.w};
    const uint32_t wy[4] = {vy.x, vy.y, vy.z, vy.w};
    if (g2) {
        const uint4 v2 = g2[i];
        const uint32_t w2[4] = {v2.x, v2.y, v2.z, v2.w};
#pragma unroll
        for (int k = 0; k < 4; k++) wg[k] = pack_bf16(bf16_lo(wg[k]) + bf16_lo(w2[k]), bf16_hi(wg[k]) + bf16_hi(w2[k]));
    }
#pragma unroll
    for (int k = 0; k < 4; k++)      // bf16 halves compared as signed 16-bit integers: +0, -0 and negatives drop
        wg[k] &= ((int16_t)(wy[k] & 0xffffu) > 0 ? 0x0000ffffu : 0u) | ((int32_t)wy[k] >= 0x10000 ? 0xffff0000u : 0u);
    gm[i] = make_uint4(wg[0], wg[1], wg[2], wg[3]);
}
int launch_add_relu(const void *a, const void *b, void *y, size_t n, hipStream_t stream) {
    if (n == 0 || n % 8) return set_error(DSRG_ERR_INVALID, "add_relu: element count must be a positive multiple of 8");
    hipLaunchKernelGGL(add_relu_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, stream, (const uint4 *)a, (const uint4 *)b,
                       (uint4 *)y, n / 8);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}
int launch_relu_mask(const void *g, const void *g2, const void *y, void *gm, size_t n, hipStream_t stream) {
    if (n == 0 || n % 8) return set_error(DSRG_ERR_INVALID, "relu_mask: element count must be a positive multiple of 8");
    hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, stream, (const uint4 *)g, (const uint4 *)g2,
                       (const uint4 *)y, (uint4 *)gm, n / 8);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

// ---- the DeepLab-v2 ASPP head (ResNet-101: four dilated 3x3 classifiers 2048 -> 21 of ONE feature map, summed) as one 1x1 product -----
// out[p][o] = sum_j W_j[o] . x[p + off_j] over the J = 36 (branch, tap) pairs: Y'[q][j O + o] = W_j[o] . x[q] is ONE 1x1 convolution of x
// with the (J O) x cin matrix of all kernels' taps (no output channel padded from 21 to a 128-wide tile, x read once), and the head's
// output gathers it: out[p][o] = bias[o] + sum_j Y'[p + off_j][j O + o] (zero outside the map; j ascending, fp32).  Backward: the
// gradient of Y' is the scatter of g, G'[q][j O + o] = g[q - off_j][o], and the data / weight gradients are the 1x1 layer's.
struct AsppShift {
    int J, O, CT;           // (branch, tap) pairs, outputs per pair, channels of Y' / G' (J O rounded up; the rest zero)
    int dy[36], dx[36];
};
__global__ __launch_bounds__(256) void aspp_shift_sum_kernel(const uint16_t *__restrict__ yp, const float *__restrict__ bias, float *__restrict__ out,
                                                             AsppShift s, int B, int H, int W) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * H * W * s.O) return;
    const size_t p = idx / s.O;
    const int o = (int)(idx - p * s.O);
    const int b = (int)(p / ((size_t)H * W)), rem = (int)(p - (size_t)b * H * W), y = rem / W, x = rem - y * W;
    float acc = bias ? bias[o] : 0.0f;
    for (int j = 0; j < s.J; j++) {
        const int yy = y + s.dy[j], xx = x + s.dx[j];
        if (yy >= 0 && yy < H && xx >= 0 && xx < W)
            acc = acc + __uint_as_float((uint32_t)yp[(((size_t)b * H + yy) * W + xx) * s.CT + j * s.O + o] << 16);
    }
    out[idx] = acc;
}
__global__ __launch_bounds__(256) void aspp_shift_gather_kernel(const float *__restrict__ g, uint4 *__restrict__ gp, AsppShift s, int B, int H, int W) {
    // eight channels (one 16-byte store) per thread
    const int C8 = s.CT >> 3;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)B * H * W * C8) return;
    const size_t p = idx / C8;
    const int c0 = (int)(idx - p * C8) << 3;
    const int b = (int)(p / ((size_t)H * W)), rem = (int)(p - (size_t)b * H * W), y = rem / W, x = rem - y * W;
    int j = c0 / s.O, o = c0 - j * s.O;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        v[e] = 0.0f;
        if (j < s.J) {
            const int yy = y - s.dy[j], xx = x - s.dx[j];
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) v[e] = g[(((size_t)b * H + yy) * W + xx) * s.O + o];
        }
        if (++o == s.O) { o = 0; j++; }
    }
    gp[idx] = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
}
static int aspp_shift_args(AsppShift &s, const int *offsets, int J, int O, int CT) {
    if (!offsets || J < 1 || J > 36 || O < 1 || CT < J * O || CT % 8) return set_error(DSRG_ERR_INVALID, "aspp shift: 1..36 (dy, dx) pairs, CT >= J * O, 8 | CT");
    memset(&s, 0, sizeof(s));
    s.J = J; s.O = O; s.CT = CT;
    for (int j = 0; j < J; j++) { s.dy[j] = offsets[2 * j]; s.dx[j] = offsets[2 * j + 1]; }
    return DSRG_OK;
}
int launch_aspp_shift_sum(const void *yp, const float *bias, float *out, const int *offsets, int J, int O, int CT, int B, int H, int W,
                          hipStream_t stream) {
    AsppShift s;
    if (int rc = aspp_shift_args(s, offsets, J, O, CT)) return rc;
    const size_t n = (size_t)B * H * W * O;
    hipLaunchKernelGGL(aspp_shift_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const uint16_t *)yp, bias, out, s, B, H, W);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}
int launch_aspp_shift_gather(const float *g, void *gp, const int *offsets, int J, int O, int CT, int B, int H, int W, hipStream_t stream) {
    AsppShift s;
    if (int rc = aspp_shift_args(s, offsets, J, O, CT)) return rc;
    const size_t n = (size_t)B * H * W * (CT / 8);
    hipLaunchKernelGGL(aspp_shift_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g, (uint4 *)gp, s, B, H, W);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

// ---- bias gradient of a (rows, C) bf16 matrix for any C <= 256 (the 21-channel fc8 outputs): column sums in f32 -------------
// lanes walk the flat array, so a wave reads 128 contiguous bytes; thread t always meets channel (t % C) because the row
// group a block advances by is a whole number of rows.  Partials per block, then bias_finalize_kernel.
__global__ __launch_bounds__(kRbThreads) void bias_grad_kernel(const unsigned short *__restrict__ g, float *__restrict__ part,
                                                               int rows, int C, int rows_per_block) {
    __shared__ float red[kRbThreads];
    const int R = kRbThreads / C;                                        // rows covered by one sweep of the block
    const int rl = threadIdx.x / C, c = threadIdx.x - rl * C;
    float acc = 0.f;
    if (rl < R) {
        const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
        for (int r = r0 + rl; r < r1; r += R) acc += __uint_as_float((uint32_t)g[(size_t)r * C + c] << 16);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if ((int)threadIdx.x < C) {
        float s = 0.f;
        for (int l = 0; l < R; ++l) s += red[l * C + threadIdx.x];
        part[(size_t)blockIdx.x * C + threadIdx.x] = s;
    }
}
int launch_bias_grad(const void *g, float *bias_grad, float *part, int part_blocks, long rows, int C, hipStream_t stream) {
    if (C < 1 || C > kRbThreads) return set_error(DSRG_ERR_UNSUPPORTED, "bias_grad: 1..256 channels");
    if (rows <= 0 || rows > 0x7fffffffL || part_blocks < 1) return set_error(DSRG_ERR_INVALID, "bias_grad: bad arguments");
    const int R = kRbThreads / C;
    int rpb = (int)((rows + part_blocks - 1) / part_blocks);
    rpb = ((rpb + R - 1) / R) * R;
    const int nblk = (int)((rows + rpb - 1) / rpb);
    hipLaunchKernelGGL(bias_grad_kernel, dim3(nblk), dim3(kRbThreads), 0, stream, (const unsigned short *)g, part, (int)rows, C, rpb);
    DSRG_LAUNCH_CHECK();
    if (!defer_reduction(1, part, bias_grad, nblk, C))
        hipLaunchKernelGGL(bias_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, stream, part, bias_grad, nblk, C);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

static int pool_check(int B, int H, int W, int OH, int OW, int C, int stride) {
    if (C % 8 != 0) return set_error(DSRG_ERR_UNSUPPORTED, "maxpool3x3: channels must be a multiple of 8");
    if (stride != 1 && stride != 2) return set_error(DSRG_ERR_UNSUPPORTED, "maxpool3x3: stride must be 1 or 2");
    if (B <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return set_error(DSRG_ERR_INVALID, "maxpool3x3: bad shape");
    if ((OH - 1) * stride - 1 >= H || (OW - 1) * stride - 1 >= W) return set_error(DSRG_ERR_INVALID, "maxpool3x3: a window starts outside the image");
    return DSRG_OK;
}

int launch_maxpool3x3_fwd(const void *in, void *out, void *code, int B, int H, int W, int OH, int OW, int C, int stride,
                          hipStream_t stream, bool relu_in) {
    int rc = pool_check(B, H, W, OH, OW, C, stride);
    if (rc) return rc;
    const size_t total = (size_t)B * OH * OW * (C / 8);
    if (relu_in)
        hipLaunchKernelGGL(maxpool3x3_fwd_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const uint4 *)in,
                           (uint4 *)out, (uint2 *)code, B, H, W, OH, OW, C / 8, stride);
    else
        hipLaunchKernelGGL(maxpool3x3_fwd_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const uint4 *)in,
                           (uint4 *)out, (uint2 *)code, B, H, W, OH, OW, C / 8, stride);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

int launch_maxpool3x3_bwd_relu(const void *gout, const void *code, const void *y, void *gin, float *bias_grad, float *part,
                                int part_blocks, int B, int H, int W, int OH, int OW, int C, hipStream_t stream) {
    int rc = pool_check(B, H, W, OH, OW, C, 2);
    if (rc) return rc;
    const int C8 = C / 8;
    if (kRbThreads % C8 != 0) return set_error(DSRG_ERR_UNSUPPORTED, "maxpool3x3_bwd_relu: channels / 8 must divide %d", kRbThreads);
    if (part_blocks < 1) return set_error(DSRG_ERR_INVALID, "maxpool3x3_bwd_relu: no partial-sum blocks");
    const size_t total = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2) * C8;
    size_t ipb = (total + part_blocks - 1) / part_blocks;
    ipb = (ipb + kRbThreads - 1) / kRbThreads * kRbThreads;                  // whole rounds: a thread keeps its channel group
    const int nblk = (int)((total + ipb - 1) / ipb);
    if (y)
        hipLaunchKernelGGL(maxpool3x3_s2_bwd_relu_kernel<true>, dim3(nblk), dim3(kRbThreads), 0, stream, (const uint4 *)gout,
                           (const uint2 *)code, (const uint4 *)y, (uint4 *)gin, part, B, H, W, OH, OW, C8, ipb);
    else
        hipLaunchKernelGGL(maxpool3x3_s2_bwd_relu_kernel<false>, dim3(nblk), dim3(kRbThreads), 0, stream, (const uint4 *)gout,
                           (const uint2 *)code, (const uint4 *)nullptr, (uint4 *)gin, part, B, H, W, OH, OW, C8, ipb);
    DSRG_LAUNCH_CHECK();
    if (!defer_reduction(1, part, bias_grad, nblk, C))
        hipLaunchKernelGGL(bias_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, stream, part, bias_grad, nblk, C);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

int launch_maxpool3x3_bwd(const void *gout, const void *code, void *gin, int B, int H, int W, int OH, int OW, int C,
                          int stride, hipStream_t stream) {
    int rc = pool_check(B, H, W, OH, OW, C, stride);
    if (rc) return rc;
    if (stride == 2) {
        const size_t blocks2 = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
        hipLaunchKernelGGL(maxpool3x3_s2_bwd_kernel, dim3((unsigned)((blocks2 + 255) / 256)), dim3(256), 0, stream,
                           (const uint4 *)gout, (const uint2 *)code, (uint4 *)gin, B, H, W, OH, OW, C / 8);
        DSRG_LAUNCH_CHECK();
        return DSRG_OK;
    }
    const size_t total = (size_t)B * H * W * (C / 8);
    hipLaunchKernelGGL(maxpool3x3_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const uint4 *)gout,
                       (const uint2 *)code, (uint4 *)gin, B, H, W, OH, OW, C / 8, stride);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}



// ---------------------------------------------------------------------------------
// fc8-SEC heads: the four 1x1 classifiers fc8-SEC_k and their Eltwise SUM (train-s.prototxt:461-744) with bf16
// activations, FLOAT32 weights, float32 accumulation and a float32 NCHW result — the scores that feed Softmax + 1e-4, the
// CRF and the 0.85 / 0.99 region-growing thresholds never pass through bf16.  bf16 -> f32 is exact, so the forward equals
// the fp32 convolution of the (bf16-valued) fc7 outputs.  Skinny GEMMs (21 outputs): forward and weight gradient on the
// f32-input MFMA (v_mfma_f32_32x32x2_f32: exact f32 fma chains at the f32 vector rate, operands straight from global
// memory, no LDS staging), the data gradient on the VALU (output-bandwidth-bound).
//   MFMA 32x32x2 operand maps (cdna_hip_programming.md §3): A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31],
//   C/D: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kHeadOutPad = 32;          // MFMA tile width; outputs beyond O are computed on clamped rows and never stored
struct HeadArgs {
    const uint16_t *x[4];      // NBR activations, (M, K) bf16 row-major (NHWC)
    const float *w;            // (NBR, O, K)
    const float *bias;         // (NBR, O) or nullptr
    float *out;                // (B, O, HW)
    int nbr, M, K, O, HW;
};
__device__ __forceinline__ void bf16x8_to_f32(const uint4 &v, float (&f)[8]) {
    f[0] = bf16_lo(v.x); f[1] = bf16_hi(v.x); f[2] = bf16_lo(v.y); f[3] = bf16_hi(v.y);
    f[4] = bf16_lo(v.z); f[5] = bf16_hi(v.z); f[6] = bf16_lo(v.w); f[7] = bf16_hi(v.w);
}
}  // namespace

// one workgroup = one 32-row tile; wave k = branch k (its whole K range), partial tiles summed through LDS in branch order
// (the reference's Eltwise SUM order, each with its own bias).  Per 8 MFMAs a lane loads 16 B of x and 32 B of W.
__global__ __launch_bounds__(256) void heads_fwd_kernel(HeadArgs a) {
    __shared__ float part[4][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.x * 32;
    const int row = lane & 31, half = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    if (wave < a.nbr) {
        const int mr = min(m0 + row, a.M - 1);                      // clamped rows are never stored
        const uint16_t *xp = a.x[wave] + (size_t)mr * a.K + half * 8;
        const float *wp = a.w + ((size_t)wave * a.O + min(row, a.O - 1)) * a.K + half * 8;
        // two 16-channel steps in flight ahead of the MFMAs that consume them (K % 256 == 0)
        uint4 xv[2];
        float4 wv[2][2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            xv[u] = *reinterpret_cast<const uint4 *>(xp + u * 16);
            wv[u][0] = *reinterpret_cast<const float4 *>(wp + u * 16);
            wv[u][1] = *reinterpret_cast<const float4 *>(wp + u * 16 + 4);
        }
        for (int kb = 0; kb < a.K; kb += 32) {
#pragma unroll
            for (int u = 0; u < 2; u++) {
                float xf[8];
                bf16x8_to_f32(xv[u], xf);
                const float wf[8] = {wv[u][0].x, wv[u][0].y, wv[u][0].z, wv[u][0].w, wv[u][1].x, wv[u][1].y, wv[u][1].z, wv[u][1].w};
                const int kn = kb + 32 + u * 16;
                if (kn < a.K) {
                    xv[u] = *reinterpret_cast<const uint4 *>(xp + kn);
                    wv[u][0] = *reinterpret_cast<const float4 *>(wp + kn);
                    wv[u][1] = *reinterpret_cast<const float4 *>(wp + kn + 4);
                }
#pragma unroll
                for (int j = 0; j < 8; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[j], wf[j], acc, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; r++) part[wave][r][lane] = acc[r];
    __syncthreads();
    // thread t: output column t / 8, rows 4 (t % 8) .. + 3 of the tile
    const int col = threadIdx.x >> 3, r4 = (threadIdx.x & 7) * 4;
    if (col < a.O) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int rr = r4 + q, m = m0 + rr;
            if (m >= a.M) break;
            const int hf = (rr >> 2) & 1, reg = (rr & 3) + 4 * (rr >> 3), ln = col + 32 * hf;
            float s = 0.0f;
            for (int k = 0; k < a.nbr; k++) {
                const float sk = part[k][reg][ln] + (a.bias ? a.bias[k * a.O + col] : 0.0f);
                s = k == 0 ? sk : s + sk;
            }
            const int b = m / a.HW, hw = m - b * a.HW;
            a.out[((size_t)b * a.O + col) * a.HW + hw] = s;
        }
    }
}

// data gradient: gx_k[m][c] = sum_o g[m][o] W_k[o][c] -> bf16 (M, K) row-major.  One workgroup = 16 T rows of one branch (T
// tiles of 16 in turn), thread = 8 channels x 8 rows; g tile through LDS (broadcast reads), W rows straight from L2.
// MASK: the backward of the ReLU (+ Dropout) that produced x_k rides in the store — values kept where x_k > 0 (times
// `scale`), and the column sums of what was stored (that layer's bias gradient) go out as one partial row per workgroup.
struct HeadMask {
    const uint16_t *y[4];
    float *part;            // (nbr, gridDim.x, K)
    float scale;
    int tiles;              // 16-row tiles per workgroup
};
template <bool MASK>
__global__ __launch_bounds__(256) void heads_bwd_dx_kernel(const float *__restrict__ g, const float *__restrict__ w,
                                                           uint4 *__restrict__ gx, int M, int K, int O, int HW, size_t branch_stride,
                                                           HeadMask hm) {
    __shared__ __attribute__((aligned(16))) float gs[32][16];        // [o][row]
    __shared__ __attribute__((aligned(16))) float cs_l[MASK ? 128 : 1][8];
    const int k = blockIdx.y, t = threadIdx.x;
    const int K8 = K / 8, T = MASK ? hm.tiles : 1;
    uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<unsigned char *>(gx) + (size_t)k * branch_stride);
    for (int cg0 = 0; cg0 < K8; cg0 += 128) {                        // workgroup-uniform trip counts (barriers inside)
        const int cg = cg0 + (t & 127), rh = t >> 7;                 // rows 8 rh .. 8 rh + 7 of a tile
        const bool live = cg < K8;
        float cs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int ti = 0; ti < T; ti++) {
            const int m0 = (blockIdx.x * T + ti) * 16;
            if (m0 >= M) break;
            if (MASK || cg0 > 0) __syncthreads();                    // the previous tile's readers are done with gs
            for (int e = t; e < 32 * 16; e += 256) {
                const int o = e >> 4, r = e & 15, m = m0 + r;
                float v = 0.0f;
                if (o < O && m < M) { const int b = m / HW, hw = m - b * HW; v = g[((size_t)b * O + o) * HW + hw]; }
                gs[o][r] = v;
            }
            __syncthreads();
            if (!live) continue;
            float acc[8][8];
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int c = 0; c < 8; c++) acc[r][c] = 0.0f;
            const float *wk = w + (size_t)k * O * K + (size_t)cg * 8;
#pragma unroll 7
            for (int o = 0; o < O; o++) {
                const float4 w0 = *reinterpret_cast<const float4 *>(wk + (size_t)o * K), w1 = *reinterpret_cast<const float4 *>(wk + (size_t)o * K + 4);
                const float4 g0 = *reinterpret_cast<const float4 *>(&gs[o][rh * 8]), g1 = *reinterpret_cast<const float4 *>(&gs[o][rh * 8 + 4]);
                const float wf[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                const float gf[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                for (int r = 0; r < 8; r++)
#pragma unroll
                    for (int c = 0; c < 8; c++) acc[r][c] = __builtin_fmaf(gf[r], wf[c], acc[r][c]);
            }
            // (mask rows fetched four at a time behind the fma loop: holding all eight through it costs the kernel half its waves)
            uint4 yk[4];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int m = m0 + rh * 8 + r;
                if (MASK && (r & 3) == 0) {
#pragma unroll
                    for (int r2 = 0; r2 < 4; r2++) {
                        const int m2 = m + r2;
                        yk[r2] = m2 < M ? *reinterpret_cast<const uint4 *>(hm.y[k] + (size_t)m2 * K + (size_t)cg * 8) : make_uint4(0, 0, 0, 0);
                    }
                }
                uint32_t o4[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (MASK) {
                        const uint4 yr = yk[r & 3];
                        const uint32_t y = e == 0 ? yr.x : e == 1 ? yr.y : e == 2 ? yr.z : yr.w;
                        const uint32_t keep = ((int16_t)(y & 0xffffu) > 0 ? 0x0000ffffu : 0u) | ((int32_t)y >= 0x10000 ? 0xffff0000u : 0u);
                        o4[e] = pack_bf16(acc[r][2 * e] * hm.scale, acc[r][2 * e + 1] * hm.scale) & keep;
                        cs[2 * e] += __uint_as_float(o4[e] << 16);                 // (rows past the end: mask 0)
                        cs[2 * e + 1] += __uint_as_float(o4[e] & 0xffff0000u);
                    } else {
                        o4[e] = pack_bf16(acc[r][2 * e], acc[r][2 * e + 1]);
                    }
                }
                if (m < M) dst[(size_t)m * K8 + cg] = make_uint4(o4[0], o4[1], o4[2], o4[3]);
            }
        }
        if (MASK) {                                                  // the two row halves of the tiles, then one partial row
            __syncthreads();
            if (rh == 1 && live) {
                *reinterpret_cast<float4 *>(&cs_l[t & 127][0]) = make_float4(cs[0], cs[1], cs[2], cs[3]);
                *reinterpret_cast<float4 *>(&cs_l[t & 127][4]) = make_float4(cs[4], cs[5], cs[6], cs[7]);
            }
            __syncthreads();
            if (rh == 0 && live) {
                float *p = hm.part + ((size_t)k * gridDim.x + blockIdx.x) * K + (size_t)cg * 8;
                const float4 a0 = *reinterpret_cast<const float4 *>(&cs_l[t][0]), a1 = *reinterpret_cast<const float4 *>(&cs_l[t][4]);
                *reinterpret_cast<float4 *>(p) = make_float4(cs[0] + a0.x, cs[1] + a0.y, cs[2] + a0.z, cs[3] + a0.w);
                *reinterpret_cast<float4 *>(p + 4) = make_float4(cs[4] + a1.x, cs[5] + a1.y, cs[6] + a1.z, cs[7] + a1.w);
            }
        }
    }
}

// The same data gradient for the shape the net has (O == 21 classes, K a multiple of 1024), bound by its 8 bytes per element (the
// bf16 store + the bf16 mask read) instead of by the latency chain of the tiled form above: one workgroup = one branch, a STRIP of
// consecutive rows, a block of 1024 channels; thread t keeps W_k[0..O)[4 t .. 4 t + 3] in registers for the whole strip (84
// VGPRs), a row's O gradient values are workgroup-uniform (scalar loads: the row index is a function of the block and the loop
// counter only), so a row costs O x 4 fmas per thread, one 8-byte mask load and one 8-byte store per thread — a wave reads and
// writes 512 contiguous bytes, a workgroup the row's whole 2 KB.  Four rows per iteration are in flight.  Same fma order over the
// outputs as heads_bwd_dx_kernel: identical bits.  Column sums: a thread adds up what it stored, one partial row per workgroup.
template <int O, bool MASK>
__global__ __launch_bounds__(256) void heads_bwd_dx_rows_kernel(const float *__restrict__ g, const float *__restrict__ w,
                                                                unsigned char *__restrict__ gx, int M, int K, int HW, size_t branch_stride,
                                                                HeadMask hm, int rows_per_wg) {
    constexpr int OP = (O + 3) & ~3;                                 // a row of the strip's gradient in LDS: O floats padded to 16 bytes
    extern __shared__ __attribute__((aligned(16))) float gs_rows[];  // [rows_per_wg][OP]
    const int k = blockIdx.y, t = threadIdx.x;
    const int c = blockIdx.z * 1024 + 4 * t;
    const int r0 = blockIdx.x * rows_per_wg, r1 = min(M, r0 + rows_per_wg), nr = r1 - r0;
    for (int e = t; e < nr * OP; e += 256) {                         // (NCHW gradient: for one output, consecutive rows are contiguous)
        const int o = e / nr, r = e - o * nr, m = r0 + r;
        float v = 0.0f;
        if (o < O) { const int b = m / HW, hw = m - b * HW; v = g[((size_t)b * O + o) * HW + hw]; }
        gs_rows[r * OP + o] = v;
    }
    float wr[O][4];
#pragma unroll
    for (int o = 0; o < O; o++) {
        const float4 v = *reinterpret_cast<const float4 *>(w + ((size_t)k * O + o) * K + c);
        wr[o][0] = v.x; wr[o][1] = v.y; wr[o][2] = v.z; wr[o][3] = v.w;
    }
    __syncthreads();
    unsigned char *dst = gx + (size_t)k * branch_stride;
    const uint16_t *yk = MASK ? hm.y[k] : nullptr;
    float cs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int R = 4;
    // the mask rows of iteration i + 1 are on their way while iteration i computes (8 x 8 bytes per lane in flight: what stands
    // between this kernel and its bandwidth is the latency of those loads, nothing else reads HBM here)
    uint2 yn[R];
    auto fetch_masks = [&](int r) {
#pragma unroll
        for (int j = 0; j < R; j++) yn[j] = r + j < nr ? *reinterpret_cast<const uint2 *>(yk + (size_t)(r0 + r + j) * K + c) : make_uint2(0u, 0u);
    };
    if (MASK) fetch_masks(0);
    for (int r = 0; r < nr; r += R) {
        uint2 y[R];
        if (MASK) {
#pragma unroll
            for (int j = 0; j < R; j++) y[j] = yn[j];
            if (r + R < nr) fetch_masks(r + R);
        }
        float acc[R][4];
#pragma unroll
        for (int j = 0; j < R; j++) {
            const float *gr = gs_rows + min(r + j, nr - 1) * OP;    // (uniform address: every lane reads the same words — a broadcast)
            float gv[OP];
#pragma unroll
            for (int q = 0; q < OP / 4; q++) {
                const float4 v = *reinterpret_cast<const float4 *>(gr + 4 * q);
                gv[4 * q] = v.x; gv[4 * q + 1] = v.y; gv[4 * q + 2] = v.z; gv[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int e = 0; e < 4; e++) acc[j][e] = 0.0f;
#pragma unroll
            for (int o = 0; o < O; o++)
#pragma unroll
                for (int e = 0; e < 4; e++) acc[j][e] = __builtin_fmaf(gv[o], wr[o][e], acc[j][e]);
        }
#pragma unroll
        for (int j = 0; j < R; j++) {
            uint32_t o2[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                if (MASK) {
                    const uint32_t yy = e == 0 ? y[j].x : y[j].y;
                    const uint32_t keep = ((int16_t)(yy & 0xffffu) > 0 ? 0x0000ffffu : 0u) | ((int32_t)yy >= 0x10000 ? 0xffff0000u : 0u);
                    o2[e] = pack_bf16(acc[j][2 * e] * hm.scale, acc[j][2 * e + 1] * hm.scale) & keep;
                    cs[2 * e] += __uint_as_float(o2[e] << 16);                     // (rows past the strip: mask 0)
                    cs[2 * e + 1] += __uint_as_float(o2[e] & 0xffff0000u);
                } else {
                    o2[e] = pack_bf16(acc[j][2 * e], acc[j][2 * e + 1]);
                }
            }
            if (r + j < nr) *reinterpret_cast<uint2 *>(dst + ((size_t)(r0 + r + j) * K + c) * 2) = make_uint2(o2[0], o2[1]);
        }
    }
    if (MASK) *reinterpret_cast<float4 *>(hm.part + ((size_t)k * gridDim.x + blockIdx.x) * K + c) = make_float4(cs[0], cs[1], cs[2], cs[3]);
}

// weight gradient, stage 1: partial[rc][k][o][c] = sum over the rows of chunk rc of g[m][o] x_k[m][c].  One workgroup = one
// (branch, row chunk); wave q = channels [256 q, 256 q + 256): 8 MFMA tiles, tile t holding channels c0 + 8 j + t so that a lane's
// 16-byte load of x feeds all eight.  g^T chunk through LDS.
constexpr int kDwRows = 64;           // rows staged per LDS refill
__global__ __launch_bounds__(256) void heads_bwd_dw_kernel(HeadArgs a, const float *__restrict__ g, float *__restrict__ partial,
                                                           int rows_per_chunk) {
    __shared__ float gs[kDwRows][33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = blockIdx.y, rc = blockIdx.x;
    const int mbeg = rc * rows_per_chunk, mend = min(a.M, mbeg + rows_per_chunk);
    const int j = lane & 31, half = lane >> 5;
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;
    const int nq = a.K / 256;                                       // channel quarters handled by the waves in turn
    for (int q0 = 0; q0 < nq; q0 += 4) {                            // workgroup-uniform trip count (barriers inside)
        const int q = q0 + wave;
        const bool live = q < nq;
        const uint16_t *xq = a.x[k] + (size_t)(live ? q : 0) * 256 + j * 8;
        for (int mb = mbeg; mb < mend; mb += kDwRows) {
            __syncthreads();
            for (int e = threadIdx.x; e < kDwRows * 32; e += 256) {
                const int o = e / kDwRows, r = e % kDwRows, m = mb + r;      // lanes along the rows: g is (B, O, HW), hw fastest
                float v = 0.0f;
                if (o < a.O && m < mend) { const int b = m / a.HW, hw = m - b * a.HW; v = g[((size_t)b * a.O + o) * a.HW + hw]; }
                gs[r][o] = v;
            }
            __syncthreads();
            // the 16-byte loads of 8 row pairs are in flight ahead of the 64 MFMAs that consume them; rows past the chunk
            // read row mend - 1 (in range) and meet a zero in gs
            if (live) {
                constexpr int G = 8;
                uint4 xv[G];
                auto fetch = [&](int r2base) {
#pragma unroll
                    for (int u = 0; u < G; u++) {
                        const int m = min(mb + r2base + 2 * u + half, mend - 1);
                        xv[u] = *reinterpret_cast<const uint4 *>(xq + (size_t)m * a.K);
                    }
                };
                fetch(0);
                for (int r2 = 0; r2 < kDwRows; r2 += 2 * G) {
                    float xf[G][8], gv[G];
#pragma unroll
                    for (int u = 0; u < G; u++) {
                        bf16x8_to_f32(xv[u], xf[u]);
                        gv[u] = gs[r2 + 2 * u + half][j];
                    }
                    if (r2 + 2 * G < kDwRows) fetch(r2 + 2 * G);
#pragma unroll
                    for (int u = 0; u < G; u++)
#pragma unroll
                        for (int t = 0; t < 8; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv[u], xf[u][t], acc[t], 0, 0, 0);
                }
            }
        }
        // tile t, reg r, lane: o = (r & 3) + 8 (r >> 2) + 4 half, channel = 256 q + 8 j + t  -> 32 contiguous bytes per (lane, r)
        float *pp = partial + (((size_t)rc * a.nbr + k) * a.O) * a.K + (size_t)(live ? q : 0) * 256 + j * 8;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int o = (r & 3) + 8 * (r >> 2) + 4 * half;
            if (live && o < a.O) {
                *reinterpret_cast<float4 *>(pp + (size_t)o * a.K) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
                *reinterpret_cast<float4 *>(pp + (size_t)o * a.K + 4) = make_float4(acc[4][r], acc[5][r], acc[6][r], acc[7][r]);
            }
        }
#pragma unroll
        for (int t = 0; t < 8; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;
    }
}
// stage 2: fixed-order sum over the row chunks
__global__ void heads_bwd_dw_reduce_kernel(const float *__restrict__ partial, float *__restrict__ gw, int nchunks, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.0f;
    for (int c = 0; c < nchunks; c++) s += partial[(size_t)c * n + i];
    gw[i] = s;
}

// ---- the same two GEMMs on the bf16 MFMA with the float32 operand split into three bf16 terms -------------------------------
// f = t0 + t1 + t2 with t0 = bf16(f), t1 = bf16(f - t0), t2 = bf16(f - t0 - t1): exact for normal f (3 x 8 significant bits),
// each product with a bf16 activation is exact in f32 and the MFMA accumulates in f32 — float32 weights / gradients without
// the 16-pass f32 MFMA: three 8-pass v_mfma_f32_32x32x16_bf16 cover 16 channels where 32x32x2_f32 needs eight 16-pass ones.
namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 hbf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 hbf16x2;
typedef __attribute__((ext_vector_type(2))) float hf32x2;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 hbf16x4v;
#define DSRG_LDS_AS __attribute__((address_space(3)))
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {          // v_cvt_pk_bf16_f32, round to nearest even
    hf32x2 v = {lo, hi};
    hbf16x2 b = __builtin_convertvector(v, hbf16x2);
    return *reinterpret_cast<uint32_t *>(&b);
}
__device__ __forceinline__ void split3(const float (&f)[8], hbf16x8 &t0, hbf16x8 &t1, hbf16x8 &t2) {
    uint32_t p0[4], p1[4], p2[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float a = f[2 * i], b = f[2 * i + 1];
        p0[i] = cvt_pk_bf16(a, b);
        a -= bf16_lo(p0[i]); b -= bf16_hi(p0[i]);                              // exact
        p1[i] = cvt_pk_bf16(a, b);
        a -= bf16_lo(p1[i]); b -= bf16_hi(p1[i]);
        p2[i] = cvt_pk_bf16(a, b);
    }
    uint4 q0 = make_uint4(p0[0], p0[1], p0[2], p0[3]), q1 = make_uint4(p1[0], p1[1], p1[2], p1[3]), q2 = make_uint4(p2[0], p2[1], p2[2], p2[3]);
    t0 = *reinterpret_cast<hbf16x8 *>(&q0); t1 = *reinterpret_cast<hbf16x8 *>(&q1); t2 = *reinterpret_cast<hbf16x8 *>(&q2);
}
__device__ __forceinline__ hbf16x8 tr_frag16(DSRG_LDS_AS unsigned char *p, int stride4) {   // see conv_direct.hip
    hbf16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(reinterpret_cast<DSRG_LDS_AS hbf16x4v *>(p));
    hbf16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(reinterpret_cast<DSRG_LDS_AS hbf16x4v *>(p + stride4));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
}  // namespace

// forward: as heads_fwd_kernel (wave = branch), the x fragment is the lane's 16-byte load as it is, the weight fragment its 32-byte
// load split in three.  One workgroup = R 32-row tiles: the weight fragment (344 KB of W per workgroup, from L2) and its split are
// shared by the R tiles — R = 2 halves both per row (round 6: 97 -> see profiles/r06_fused_backward_probe.txt); the rows' arithmetic
// is the same either way (bit-identical outputs).
template <int R>
__global__ __launch_bounds__(256) void heads_fwd_split_kernel(HeadArgs a) {
    __shared__ float part[R][4][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.x * (32 * R);
    const int row = lane & 31, half = lane >> 5;
    f32x16 acc[R][3];
#pragma unroll
    for (int t = 0; t < R; t++)
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][q][r] = 0.0f;
    if (wave < a.nbr) {
        const uint16_t *xp[R];
#pragma unroll
        for (int t = 0; t < R; t++) xp[t] = a.x[wave] + (size_t)min(m0 + 32 * t + row, a.M - 1) * a.K + half * 8;      // clamped rows are never stored
        const float *wp = a.w + ((size_t)wave * a.O + min(row, a.O - 1)) * a.K + half * 8;
        constexpr int G = 4;                                        // 16-channel steps in flight (K % 256 == 0)
        uint4 xv[G][R];
        float4 wv[G][2];
#pragma unroll
        for (int u = 0; u < G; u++) {
#pragma unroll
            for (int t = 0; t < R; t++) xv[u][t] = *reinterpret_cast<const uint4 *>(xp[t] + u * 16);
            wv[u][0] = *reinterpret_cast<const float4 *>(wp + u * 16);
            wv[u][1] = *reinterpret_cast<const float4 *>(wp + u * 16 + 4);
        }
        for (int kb = 0; kb < a.K; kb += 16 * G) {
#pragma unroll
            for (int u = 0; u < G; u++) {
                hbf16x8 xa[R];
#pragma unroll
                for (int t = 0; t < R; t++) xa[t] = *reinterpret_cast<hbf16x8 *>(&xv[u][t]);
                const float wf[8] = {wv[u][0].x, wv[u][0].y, wv[u][0].z, wv[u][0].w, wv[u][1].x, wv[u][1].y, wv[u][1].z, wv[u][1].w};
                const int kn = kb + 16 * G + u * 16;
                if (kn < a.K) {
#pragma unroll
                    for (int t = 0; t < R; t++) xv[u][t] = *reinterpret_cast<const uint4 *>(xp[t] + kn);
                    wv[u][0] = *reinterpret_cast<const float4 *>(wp + kn);
                    wv[u][1] = *reinterpret_cast<const float4 *>(wp + kn + 4);
                }
                hbf16x8 w0, w1, w2;
                split3(wf, w0, w1, w2);
#pragma unroll
                for (int t = 0; t < R; t++) {
                    acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[t], w0, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[t], w1, acc[t][1], 0, 0, 0);
                    acc[t][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[t], w2, acc[t][2], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < R; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) part[t][wave][r][lane] = acc[t][0][r] + (acc[t][1][r] + acc[t][2][r]);
    __syncthreads();
    const int col = threadIdx.x >> 3, r4 = (threadIdx.x & 7) * 4;
    if (col < a.O) {
#pragma unroll
        for (int t = 0; t < R; t++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int rr = r4 + q, m = m0 + 32 * t + rr;
                if (m >= a.M) break;
                const int hf = (rr >> 2) & 1, reg = (rr & 3) + 4 * (rr >> 3), ln = col + 32 * hf;
                float s = 0.0f;
                for (int k = 0; k < a.nbr; k++) {
                    const float sk = part[t][k][reg][ln] + (a.bias ? a.bias[k * a.O + col] : 0.0f);
                    s = k == 0 ? sk : s + sk;
                }
                const int b = m / a.HW, hw = m - b * a.HW;
                a.out[((size_t)b * a.O + col) * a.HW + hw] = s;
            }
    }
}

// weight gradient, stage 1 (same partial layout [rc][k][o][c]): 32 rows of x_k (all K channels, NHWC as it lies in memory, row
// stride K * 2 + 64 bytes) and of g^T per refill in LDS; per 16-row k-step a lane takes its 8 values of g (split in three) and
// eight x fragments by transposing reads (ds_read_b64_tr_b16), wave q = channels [256 q, 256 q + 256) = 8 accumulator tiles.
constexpr int kDwRows2 = 32, kDwGPitch = 36;
__global__ __launch_bounds__(256, 2) void heads_bwd_dw_split_kernel(HeadArgs a, const float *__restrict__ g, float *__restrict__ partial,
                                                                 int rows_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dw_lds[];
    const int xstride = a.K * 2 + 64, K8 = a.K / 8;
    unsigned char *xt = dw_lds;
    float *gs = reinterpret_cast<float *>(dw_lds + kDwRows2 * xstride);          // [o][row], pitch 36 floats
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = blockIdx.y, rc = blockIdx.x;
    const int mbeg = rc * rows_per_chunk, mend = min(a.M, mbeg + rows_per_chunk);
    const int i16 = lane & 15, gq = lane >> 4;
    const int rowsel = (gq >> 1) * 8 + (i16 >> 2), colsel = (gq & 1) * 16 + 4 * (i16 & 3);
    const int nq = a.K / 256;
    const uint16_t *xk = a.x[k];
    for (int q0 = 0; q0 < nq; q0 += 4) {                            // workgroup-uniform trip count (barriers inside)
        const int q = q0 + wave;
        const bool live = q < nq;
        f32x16 acc[8];
#pragma unroll
        for (int t = 0; t < 8; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;
        DSRG_LDS_AS unsigned char *xb = (DSRG_LDS_AS unsigned char *)xt + rowsel * xstride + ((live ? q : 0) * 256 + colsel) * 2;
        for (int mb = mbeg; mb < mend; mb += kDwRows2) {
            __syncthreads();                                        // the previous refill has been consumed
            for (int e = threadIdx.x; e < kDwRows2 * 32; e += 256) {
                const int o = e / kDwRows2, r = e % kDwRows2, m = mb + r;        // lanes along the rows: g is (B, O, HW), hw fastest
                float v = 0.0f;                                     // rows past the chunk and outputs past O add nothing
                if (o < a.O && m < mend) { const int b = m / a.HW, hw = m - b * a.HW; v = g[((size_t)b * a.O + o) * a.HW + hw]; }
                gs[o * kDwGPitch + r] = v;
            }
#pragma unroll 8
            for (int v = threadIdx.x; v < kDwRows2 * K8; v += 256) {
                const int r = v / K8, cg = v - r * K8, m = min(mb + r, mend - 1);   // rows past the chunk: any row in range, g is 0 there
                *reinterpret_cast<uint4 *>(xt + r * xstride + cg * 16) = *reinterpret_cast<const uint4 *>(xk + (size_t)m * a.K + cg * 8);
            }
            __syncthreads();
            if (live) {
#pragma unroll
                for (int s2 = 0; s2 < kDwRows2 / 16; s2++) {
                    const float *gp = gs + (lane & 31) * kDwGPitch + s2 * 16 + (lane >> 5) * 8;
                    const float4 g0 = *reinterpret_cast<const float4 *>(gp), g1 = *reinterpret_cast<const float4 *>(gp + 4);
                    const float gf[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                    hbf16x8 a0, a1, a2, bx[8];
                    split3(gf, a0, a1, a2);
#pragma unroll
                    for (int t = 0; t < 8; t++) bx[t] = tr_frag16(xb + s2 * 16 * xstride + t * 64, 4 * xstride);
#pragma unroll
                    for (int t = 0; t < 8; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bx[t], acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 8; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bx[t], acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 8; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bx[t], acc[t], 0, 0, 0);
                }
            }
        }
        // tile t, reg r, lane: o = (r & 3) + 8 (r >> 2) + 4 (lane / 32), channel = 256 q + 32 t + lane % 32
        if (live) {
            float *pp = partial + (((size_t)rc * a.nbr + k) * a.O) * a.K + (size_t)q * 256 + (lane & 31);
#pragma unroll
            for (int t = 0; t < 8; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int o = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (o < a.O) pp[(size_t)o * a.K + t * 32] = acc[t][r];
                }
        }
    }
}

// DSRG_HEADS_MFMA=f32 selects the round-2a kernels on the f32-input MFMA (exact fma chains); default: the bf16 MFMA on split operands
static bool heads_f32_mfma() {
    static const bool f32 = [] { const char *e = getenv("DSRG_HEADS_MFMA"); return e && !strcmp(e, "f32"); }();
    return f32;
}
static int heads_check(int nbr, int K, int O) {
    if (nbr < 1 || nbr > 4 || O < 1 || O > kHeadOutPad || K < 256 || K % 256 != 0)
        return set_error(DSRG_ERR_UNSUPPORTED, "heads: 1..4 branches, <= %d outputs, K a multiple of 256", kHeadOutPad);
    return DSRG_OK;
}

int launch_heads_fwd(const void *const *x, int nbr, const float *w, const float *bias, float *out, int B, int HW, int K, int O,
                     hipStream_t stream) {
    int rc = heads_check(nbr, K, O);
    if (rc) return rc;
    HeadArgs a;
    for (int k = 0; k < 4; k++) a.x[k] = static_cast<const uint16_t *>(x[k < nbr ? k : 0]);
    a.w = w; a.bias = bias; a.out = out; a.nbr = nbr; a.M = B * HW; a.K = K; a.O = O; a.HW = HW;
    if (heads_f32_mfma())
        hipLaunchKernelGGL(heads_fwd_kernel, dim3((a.M + 31) / 32), dim3(256), 0, stream, a);
    else
    {
        static const int tiles = [] { const char *e = getenv("DSRG_HEAD_FWD_TILES"); const int v = e ? atoi(e) : 2; return v >= 1 && v <= 4 ? v : 2; }();      // tools: A/B
        if (tiles == 4) hipLaunchKernelGGL(heads_fwd_split_kernel<4>, dim3((a.M + 127) / 128), dim3(256), 0, stream, a);
        else if (tiles == 3) hipLaunchKernelGGL(heads_fwd_split_kernel<3>, dim3((a.M + 95) / 96), dim3(256), 0, stream, a);
        else if (tiles == 2) hipLaunchKernelGGL(heads_fwd_split_kernel<2>, dim3((a.M + 63) / 64), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(heads_fwd_split_kernel<1>, dim3((a.M + 31) / 32), dim3(256), 0, stream, a);
    }
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

// row chunks of the weight gradient: ~256 rows each so that two workgroups share a CU (one wave per SIMD each) and one's
// loads hide behind the other's MFMAs; capped so that the partial sums stay small (chunks x n x O x K floats)
int heads_bwd_chunks(int M) { int c = (M + 255) / 256; return c < 1 ? 1 : (c > 128 ? 128 : c); }

// 16-row tiles per workgroup of the masked data gradient (one partial bias row each): fewer rows per workgroup = more of them
// in flight, more partial rows to write and sum (DSRG_HEAD_TILES overrides, tools only)
static int head_mask_tiles() {
    static const int t = [] { const char *e = getenv("DSRG_HEAD_TILES"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 64 ? v : 8; }();
    return t;
}
// rows per workgroup of the row-strip form (heads_bwd_dx_rows_kernel): long enough to amortise the 84 KB of W a workgroup loads,
// short enough for a few rounds of the chip (DSRG_HEAD_ROWS overrides, tools only; 0 = the tiled kernel)
static int head_strip_rows() {
    static const int r = [] { const char *e = getenv("DSRG_HEAD_ROWS"); const int v = e ? atoi(e) : -1; return v >= 0 && v <= 512 ? v : 64; }();
    return r;
}
static bool head_strips(int K, int O) { return head_strip_rows() > 0 && O == 21 && K % 1024 == 0; }
size_t heads_bwd_relu_workspace(int nbr, int M, int K) {
    const int kHeadMaskTiles = head_mask_tiles();
    size_t rows = (size_t)((M + 16 * kHeadMaskTiles - 1) / (16 * kHeadMaskTiles));
    if (head_strip_rows() > 0) rows = std::max(rows, (size_t)((M + head_strip_rows() - 1) / head_strip_rows()));
    return (size_t)nbr * rows * (size_t)K * sizeof(float);
}

int launch_heads_bwd(const void *const *x, int nbr, const float *w, const float *g, void *gx, size_t gx_branch_stride,
                     float *gw, float *partial, int B, int HW, int K, int O, hipStream_t stream, float relu_scale, float *bias_grad,
                     void *colsum_ws, size_t colsum_ws_bytes) {
    int rc = heads_check(nbr, K, O);
    if (rc) return rc;
    const int M = B * HW;
    if (gx && relu_scale > 0.0f) {
        HeadMask hm;
        for (int k = 0; k < 4; k++) hm.y[k] = static_cast<const uint16_t *>(x[k < nbr ? k : 0]);
        const int kHeadMaskTiles = head_mask_tiles();
        hm.tiles = kHeadMaskTiles; hm.scale = relu_scale;
        const int nblk = (M + 16 * kHeadMaskTiles - 1) / (16 * kHeadMaskTiles);
        if (!bias_grad || !colsum_ws || colsum_ws_bytes < heads_bwd_relu_workspace(nbr, M, K))
            return set_error(DSRG_ERR_INVALID, "heads backward: bias gradient / scratch of the absorbed ReLU missing or too small");
        hm.part = static_cast<float *>(colsum_ws);
        int nblk_s = nblk;
        if (head_strips(K, O)) {
            nblk_s = (M + head_strip_rows() - 1) / head_strip_rows();
            hipLaunchKernelGGL((heads_bwd_dx_rows_kernel<21, true>), dim3(nblk_s, nbr, K / 1024), dim3(256),
                               (size_t)head_strip_rows() * 24 * sizeof(float), stream, g, w, static_cast<unsigned char *>(gx), M, K, HW,
                               gx_branch_stride, hm, head_strip_rows());
        } else {
            hipLaunchKernelGGL(heads_bwd_dx_kernel<true>, dim3(nblk, nbr), dim3(256), 0, stream, g, w, static_cast<uint4 *>(gx), M, K, O, HW,
                               gx_branch_stride, hm);
        }
        DSRG_LAUNCH_CHECK();
        const float *parts[4];
        float *outs[4];
        for (int k = 0; k < nbr; k++) { parts[k] = hm.part + (size_t)k * nblk_s * K; outs[k] = bias_grad + (size_t)k * K; }
        if (int rc2 = launch_igemm_colsum(parts, outs, nbr, nblk_s, K, stream)) return rc2;
    } else if (gx) {
        HeadMask hm;
        memset(&hm, 0, sizeof(hm));
        if (head_strips(K, O))
            hipLaunchKernelGGL((heads_bwd_dx_rows_kernel<21, false>), dim3((M + head_strip_rows() - 1) / head_strip_rows(), nbr, K / 1024),
                               dim3(256), (size_t)head_strip_rows() * 24 * sizeof(float), stream, g, w, static_cast<unsigned char *>(gx), M, K, HW,
                               gx_branch_stride, hm, head_strip_rows());
        else
            hipLaunchKernelGGL(heads_bwd_dx_kernel<false>, dim3((M + 15) / 16, nbr), dim3(256), 0, stream, g, w, static_cast<uint4 *>(gx), M,
                               K, O, HW, gx_branch_stride, hm);
        DSRG_LAUNCH_CHECK();
    }
    if (gw) {
        HeadArgs a;
        for (int k = 0; k < 4; k++) a.x[k] = static_cast<const uint16_t *>(x[k < nbr ? k : 0]);
        a.w = w; a.bias = nullptr; a.out = nullptr; a.nbr = nbr; a.M = M; a.K = K; a.O = O; a.HW = HW;
        const int nch = heads_bwd_chunks(M), rows = ((M + nch - 1) / nch + 1) & ~1;
        const size_t lds = (size_t)kDwRows2 * (K * 2 + 64) + 32 * kDwGPitch * sizeof(float);
        if (heads_f32_mfma() || lds > 160 * 1024) {
            hipLaunchKernelGGL(heads_bwd_dw_kernel, dim3(nch, nbr), dim3(256), 0, stream, a, g, partial, rows);
        } else {
            static LdsGrant grant;
            if (int rc2 = ensure_dynamic_lds(reinterpret_cast<const void *>(&heads_bwd_dw_split_kernel), lds, grant)) return rc2;
            hipLaunchKernelGGL(heads_bwd_dw_split_kernel, dim3(nch, nbr), dim3(256), lds, stream, a, g, partial, rows);
        }
        DSRG_LAUNCH_CHECK();
        const size_t n = (size_t)nbr * O * K;
        hipLaunchKernelGGL(heads_bwd_dw_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, partial, gw, nch, n);
        DSRG_LAUNCH_CHECK();
    }
    return DSRG_OK;
}

}  // namespace dsrg
