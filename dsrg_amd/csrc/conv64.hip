// Direct 3x3 convolution for the 64 -> 64 channel layer at full resolution (conv1_2 of train-s.prototxt:65-98, 321x321:
// backbone plumbing, no reference counterpart — Caffe's Convolution layer lives in the external framework).
//
// Why it exists: at 64 channels and 1.65 M output pixels (batch 16) the im2col route would move 1.9 GB, and MIOpen's
// implicit-GEMM kernels run at ~255 TFLOP/s here (0.48 ms forward, 0.35 ms data gradient); the layer is memory-bound at
// ~0.1 ms (211 MB in, 211 MB out).  One kernel serves the forward (bias + ReLU in the epilogue) and the data gradient (the
// same convolution with the kernel flipped and its channel axes swapped, prepared by the caller).
//
// Shape of the kernel: persistent workgroups of 4 waves, each wave at one wave per SIMD with the WHOLE weight tensor in its
// registers as MFMA B fragments (2 output tiles x 36 k-steps x 8 bf16 = 288 VGPRs: the file has 512 at this occupancy), so
// LDS carries only the input: an 8 x 16 pixel output tile with its 10 x 18 halo (26 KB, 144-byte pixel stride:
// conflict-free 16-byte reads), double buffered — the next tile's halo is fetched into registers before the 72 MFMAs
// (v_mfma_f32_32x32x16_bf16) of the current one and parked in the other buffer after them.  The output tile goes back
// through LDS for 16-byte coalesced NHWC stores.
#include "common.h"

namespace dsrg {
namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int kC = 64;                       // input = output channels
constexpr int kTH = 8, kTW = 16;             // output tile (pixels): 4 waves x (2 rows x 16 columns)
constexpr int kHH = kTH + 2, kHW = kTW + 2;  // halo tile
constexpr int kPixStride = 144;              // bytes per halo pixel in LDS (128 + 16 pad)
constexpr int kBufBytes = kHH * kHW * kPixStride;        // 25 920
constexpr int kHaloVecs = kHH * kHW * (kC * 2 / 16);     // 16-byte vectors per halo tile: 1 440
constexpr int kVecPerThread = (kHaloVecs + 255) / 256;   // 6
constexpr int kOutStride = 144;              // bytes per pixel of the output tile in LDS (16-byte aligned rows, 2-way on stores)

struct Conv64Args {
    const uint16_t *x;      // (B, H, W, 64) bf16
    const uint16_t *w;      // (64 out, 3, 3, 64 in) bf16  (= a channels_last (out, in, 3, 3) tensor)
    const float *bias;      // (64) or nullptr
    uint16_t *y;            // (B, H, W, 64) bf16
    int B, H, W, relu, tiles_x, tiles_y, ntiles;
};

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    uint32_t a = __float_as_uint(lo), b = __float_as_uint(hi);
    a = (a + 0x7fffu + ((a >> 16) & 1u)) >> 16;             // round to nearest even (finite values)
    b = (b + 0x7fffu + ((b >> 16) & 1u)) >> 16;
    return a | (b << 16);
}
}  // namespace

__global__ __launch_bounds__(256) void conv3x3_c64_kernel(Conv64Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][kBufBytes];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, kgrp = lane >> 5;

    // ---- the whole kernel tensor as B fragments: B[k][n], n = output channel, k = (tap, input channel)
    bf16x8 wf[2][36];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int ks = 0; ks < 36; ks++)
            wf[nt][ks] = *reinterpret_cast<const bf16x8 *>(a.w + (size_t)(nt * 32 + m) * 576 + ks * 16 + kgrp * 8);

    float bias_r[2][4][4];                    // the bias of the 32 output channels this lane writes
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int e = 0; e < 4; e++) bias_r[nt][q][e] = a.bias ? a.bias[nt * 32 + q * 8 + kgrp * 4 + e] : 0.0f;

    auto tile_origin = [&](int t, int &b, int &y0, int &x0) {
        const int per = a.tiles_x * a.tiles_y;
        b = t / per;
        const int r = t - b * per;
        y0 = (r / a.tiles_x) * kTH;
        x0 = (r % a.tiles_x) * kTW;
    };
    // halo vector v of a tile: pixel (hy, hx) of the 10 x 18 halo, 16-byte channel group cg
    uint4 pre[kVecPerThread];
    auto fetch = [&](int t) {
        int b, y0, x0;
        tile_origin(t, b, y0, x0);
#pragma unroll
        for (int u = 0; u < kVecPerThread; u++) {
            const int v = tid + u * 256;
            uint4 val = make_uint4(0u, 0u, 0u, 0u);
            if (v < kHaloVecs) {
                const int px = v >> 3, cg = v & 7, hy = px / kHW, hx = px - hy * kHW;
                const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
                if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W)
                    val = *reinterpret_cast<const uint4 *>(a.x + (((size_t)b * a.H + yy) * a.W + xx) * kC + cg * 8);
            }
            pre[u] = val;
        }
    };
    auto park = [&](unsigned char *buf) {
#pragma unroll
        for (int u = 0; u < kVecPerThread; u++) {
            const int v = tid + u * 256;
            if (v < kHaloVecs) *reinterpret_cast<uint4 *>(buf + (v >> 3) * kPixStride + (v & 7) * 16) = pre[u];
        }
    };

    int t = blockIdx.x, cur = 0;
    if (t >= a.ntiles) return;
    fetch(t);
    park(lds[0]);
    __syncthreads();
    // this wave's 32 pixels: tile rows 2 wave, 2 wave + 1; A[m][k]: m = pixel, k = (tap, channel)
    const int ty = 2 * wave + (m >> 4), tx = m & 15;
    for (; t < a.ntiles; t += gridDim.x) {
        const int tn = t + gridDim.x;
        if (tn < a.ntiles && !(a.relu & 4)) fetch(tn);           // next halo on its way while this tile computes
        const unsigned char *in = lds[cur];
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; r++) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
        // one wave per SIMD: nothing else hides the LDS latency of the A operand, so the four 16-byte reads of tap t + 1 are
        // issued before the eight MFMAs of tap t; sched_barrier pins that order (left alone, the scheduler sinks every read
        // next to its use and the wave waits out the LDS latency 36 times per tile)
        auto a_ptr = [&](int tap) { return in + ((ty + tap / 3) * kHW + (tx + tap % 3)) * kPixStride + kgrp * 16; };
        bf16x8 ar[2][4];
#pragma unroll
        for (int cg = 0; cg < 4; cg++) ar[0][cg] = *reinterpret_cast<const bf16x8 *>(a_ptr(0) + cg * 32);
        if (!(a.relu & 8))
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            if (tap + 1 < 9) {
#pragma unroll
                for (int cg = 0; cg < 4; cg++) ar[(tap + 1) & 1][cg] = *reinterpret_cast<const bf16x8 *>(a_ptr(tap + 1) + cg * 32);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cg = 0; cg < 4; cg++) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][tap * 4 + cg], ar[tap & 1][cg], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][tap * 4 + cg], ar[tap & 1][cg], acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                         // every wave is done reading lds[cur]
        // epilogue: the product is taken transposed, C[row = output channel][col = pixel], so a lane holds runs of four
        // consecutive channels of ONE pixel (row = (reg & 3) + 8 (reg >> 2) + 4 kgrp): bias, ReLU, four bf16 = one 8-byte LDS
        // store per run into the output tile [128 pixels][64 channels] (rows of kOutStride bytes) in lds[cur]
        unsigned char *ot = lds[cur];
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int c0 = nt * 32 + q * 8 + kgrp * 4;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    v[e] = (nt ? acc1[q * 4 + e] : acc0[q * 4 + e]) + bias_r[nt][q][e];
                    if (a.relu & 1) v[e] = fmaxf(v[e], 0.0f);
                }
                *reinterpret_cast<uint2 *>(ot + (wave * 32 + m) * kOutStride + c0 * 2) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
            }
        }
        if (tn < a.ntiles && !(a.relu & 4)) park(lds[cur ^ 1]);
        __syncthreads();
        {
            int b, y0, x0;
            tile_origin(t, b, y0, x0);
#pragma unroll
            for (int u = 0; u < 4; u++) {                        // 128 pixels x 8 vectors = 1 024 stores of 16 bytes
                const int v = tid + u * 256, px = v >> 3, cg = v & 7;
                const int yy = y0 + (px >> 4), xx = x0 + (px & 15);
                if (yy < a.H && xx < a.W && !(a.relu & 2))
                    *reinterpret_cast<uint4 *>(a.y + (((size_t)b * a.H + yy) * a.W + xx) * kC + cg * 8) =
                        *reinterpret_cast<const uint4 *>(ot + px * kOutStride + cg * 16);
            }
        }
        __syncthreads();                                         // lds[cur] is free for the tile after next
        cur ^= 1;
    }
}

int launch_conv3x3_c64(const void *x, const void *w, const float *bias, void *y, int B, int H, int W, int relu,
                       hipStream_t stream) {
    Conv64Args a;
    a.x = static_cast<const uint16_t *>(x); a.w = static_cast<const uint16_t *>(w); a.bias = bias;
    a.y = static_cast<uint16_t *>(y); a.B = B; a.H = H; a.W = W; a.relu = relu;
    a.tiles_x = (W + kTW - 1) / kTW; a.tiles_y = (H + kTH - 1) / kTH;
    const long nt = (long)B * a.tiles_x * a.tiles_y;
    if (nt < 1 || nt > 0x7fffffffL) return set_error(DSRG_ERR_INVALID, "conv3x3_c64: bad shape");
    a.ntiles = (int)nt;
    static const int n_cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    const int grid = a.ntiles < n_cus ? a.ntiles : n_cus;        // persistent: one workgroup per CU
    hipLaunchKernelGGL(conv3x3_c64_kernel, dim3(grid), dim3(256), 0, stream, a);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

}  // namespace dsrg
