// Direct 3x3 convolutions for the narrow layers at the two large resolutions: 64 -> 64 at 321x321 (conv1_2 of
// train-s.prototxt:65-98) and 64 -> 128 / 128 -> 128 at 161x161 (conv2_1, conv2_2, :110-160), plus 128 -> 64 for the data
// gradient of conv2_1 (backbone plumbing, no reference counterpart — Caffe's Convolution layer lives in the external
// framework).
//
// Why they exist: these layers have few channels and many pixels (1.65 M / 0.41 M per batch of 16).  The im2col route
// moves 9x the activation (1.9 GB for conv1_2, 0.96 GB for conv2_2) and MIOpen's implicit-GEMM kernels run at ~255 TFLOP/s
// here (conv1_2: 0.48 ms forward, 0.35 ms data gradient); the layers are memory-bound at ~0.1 ms.  One kernel serves the
// forward (bias + ReLU in the epilogue) and the data gradient (the same convolution with the kernel flipped and its channel
// axes swapped, prepared by the caller).
//
// Shape of the kernel: persistent workgroups of 4 waves at one wave per SIMD, each wave with its share of the weight tensor
// in registers as MFMA fragments — 288 VGPRs in every variant (the file has 512 at this occupancy):
//     64 -> 64    every wave holds all 64 outputs (2 tiles x 36 k-steps), the 4 waves split the 128 pixels of a tile
//     64 -> 128   a wave holds 64 of the outputs, 2 x 2 waves split outputs x pixels (2 M-tiles of 32 pixels per wave)
//    128 -> 128   a wave holds 32 of the outputs (1 tile x 72 k-steps) and visits all 4 M-tiles
//    128 -> 64    a wave holds 32 of the outputs, 2 x 2 waves split outputs x pixels
// so LDS carries only the input: an 8 x 16 pixel output tile with its 10 x 18 halo (pixel stride = channels * 2 + 16 bytes:
// conflict-free 16-byte reads), double buffered — the next tile's halo is fetched into registers before the MFMAs
// (v_mfma_f32_32x32x16_bf16) of an M-tile and parked in the other buffer after them.  The output tile goes back through LDS
// for 16-byte coalesced NHWC stores.
#include "common.h"

namespace dsrg {
namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int kTH = 8, kTW = 16;             // output tile (pixels): 4 M-tiles of (2 rows x 16 columns)
constexpr int kHH = kTH + 2, kHW = kTW + 2;  // halo tile

template <int CIN, int COUT> struct Cfg {
    static constexpr int CG = CIN / 16;                    // 16-channel k-steps per tap
    static constexpr int KS = 9 * CG;                      // k-steps
    static constexpr int TPW = CIN == 64 ? 2 : 1;          // 32-channel output tiles per wave: TPW * KS * 4 = 288 VGPRs
    static constexpr int NG = COUT / (32 * TPW);           // waves across the output channels
    static constexpr int PG = 4 / NG;                      // waves across the pixels
    static constexpr int MT = 4 / PG;                      // 32-pixel M-tiles per wave
    static constexpr int SPT = CG / 4;                     // steps (4 k-steps) per tap
    static constexpr int STEPS = 9 * SPT;
    static constexpr int IN_STRIDE = CIN * 2 + 16;         // bytes per halo pixel in LDS
    static constexpr int OUT_STRIDE = COUT * 2 + 16;       // bytes per pixel of the output tile in LDS (2-way on the 8-byte stores)
    static constexpr int IN_BYTES = kHH * kHW * IN_STRIDE, OUT_BYTES = kTH * kTW * OUT_STRIDE;
    static constexpr int BUF = IN_BYTES > OUT_BYTES ? IN_BYTES : OUT_BYTES;   // a buffer is a halo tile, then an output tile
    static constexpr int VPP = CIN / 8;                    // 16-byte vectors per input pixel
    static constexpr int HALO_VECS = kHH * kHW * VPP;
    static constexpr int VPT = (HALO_VECS + 255) / 256;    // per thread: 6 / 12
    static constexpr int VPC = (VPT + MT - 1) / MT;        // per thread and chunk: the fetch is spread over the M-tiles
    static constexpr int OVP = COUT / 8;                   // 16-byte vectors per output pixel
    static constexpr int OVT = kTH * kTW * OVP / 256;      // per thread: 4 / 8
    static_assert(NG * PG == 4 && NG >= 1 && MT * PG == 4, "4 waves");
    static_assert(BUF % 16 == 0, "alignment");
};

struct ConvArgs {
    const uint16_t *x;      // (B, H, W, CIN) bf16
    const uint16_t *w;      // (COUT, 3, 3, CIN) bf16  (= a channels_last (out, in, 3, 3) tensor)
    const float *bias;      // (COUT) or nullptr
    uint16_t *y;            // (B, H, W, COUT) bf16
    int B, H, W, relu, tiles_x, tiles_y, ntiles;
};

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {   // v_cvt_pk_bf16_f32: round to nearest even
    f32x2 v = {lo, hi};
    bf16x2 b = __builtin_convertvector(v, bf16x2);
    return *reinterpret_cast<uint32_t *>(&b);
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv3x3_direct_kernel(ConvArgs a) {
    using C = Cfg<CIN, COUT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char conv_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, kgrp = lane >> 5;
    const int ng = wave % C::NG, pg = wave / C::NG;

    // ---- this wave's share of the kernel tensor as MFMA fragments: [k][n], n = output channel, k = (tap, input channel)
    bf16x8 wf[C::TPW][C::KS];
#pragma unroll
    for (int j = 0; j < C::TPW; j++)
#pragma unroll
        for (int ks = 0; ks < C::KS; ks++)
            wf[j][ks] = *reinterpret_cast<const bf16x8 *>(a.w + (size_t)((ng * C::TPW + j) * 32 + m) * (9 * CIN) + ks * 16 + kgrp * 8);

    float bias_r[C::TPW][4][4];               // the bias of the output channels this lane writes
#pragma unroll
    for (int j = 0; j < C::TPW; j++)
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int e = 0; e < 4; e++)
                bias_r[j][q][e] = a.bias ? a.bias[(ng * C::TPW + j) * 32 + q * 8 + kgrp * 4 + e] : 0.0f;

    auto tile_origin = [&](int t, int &b, int &y0, int &x0) {
        const int per = a.tiles_x * a.tiles_y;
        b = t / per;
        const int r = t - b * per;
        y0 = (r / a.tiles_x) * kTH;
        x0 = (r % a.tiles_x) * kTW;
    };
    // halo vector v of a tile: pixel (hy, hx) of the 10 x 18 halo, 16-byte channel group cg; chunk c of MT per thread
    uint4 pre[C::VPC];
    auto fetch = [&](int t, int c) {
        int b, y0, x0;
        tile_origin(t, b, y0, x0);
#pragma unroll
        for (int u = 0; u < C::VPC; u++) {
            const int v = tid + (c * C::VPC + u) * 256;
            uint4 val = make_uint4(0u, 0u, 0u, 0u);
            if (v < C::HALO_VECS) {
                const int px = v / C::VPP, cg = v % C::VPP, hy = px / kHW, hx = px - hy * kHW;
                const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
                if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W)
                    val = *reinterpret_cast<const uint4 *>(a.x + (((size_t)b * a.H + yy) * a.W + xx) * CIN + cg * 8);
            }
            pre[u] = val;
        }
    };
    auto park = [&](unsigned char *buf, int c) {
#pragma unroll
        for (int u = 0; u < C::VPC; u++) {
            const int v = tid + (c * C::VPC + u) * 256;
            if (v < C::HALO_VECS) *reinterpret_cast<uint4 *>(buf + (v / C::VPP) * C::IN_STRIDE + (v % C::VPP) * 16) = pre[u];
        }
    };

    int t = blockIdx.x, cur = 0;
    if (t >= a.ntiles) return;
#pragma unroll
    for (int c = 0; c < C::MT; c++) {
        fetch(t, c);
        park(conv_lds, c);
    }
    __syncthreads();
    for (; t < a.ntiles; t += gridDim.x) {
        const int tn = t + gridDim.x;
        const bool more = tn < a.ntiles;
        unsigned char *in = conv_lds + cur * C::BUF, *other = conv_lds + (cur ^ 1) * C::BUF;
        f32x16 acc[C::MT][C::TPW];
#pragma unroll
        for (int mt = 0; mt < C::MT; mt++) {
            if (more) fetch(tn, mt);                             // a slice of the next halo on its way while this M-tile computes
#pragma unroll
            for (int j = 0; j < C::TPW; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[mt][j][r] = 0.0f;
            // this wave's 32 pixels of the M-tile: tile rows 2 i, 2 i + 1; A[m][k]: m = pixel, k = (tap, channel)
            const int ty = 2 * (pg * C::MT + mt) + (m >> 4), tx = m & 15;
            // one wave per SIMD: nothing else hides the LDS latency of the pixel operand, so the four 16-byte reads of step
            // s + 1 are issued before the MFMAs of step s; sched_barrier pins that order (left alone, the scheduler sinks
            // every read next to its use and the wave waits out the LDS latency at every step)
            auto a_ptr = [&](int s) {
                const int tap = s / C::SPT, h = s % C::SPT;
                return in + ((ty + tap / 3) * kHW + (tx + tap % 3)) * C::IN_STRIDE + h * 128 + kgrp * 16;
            };
            bf16x8 ar[2][4];
#pragma unroll
            for (int i = 0; i < 4; i++) ar[0][i] = *reinterpret_cast<const bf16x8 *>(a_ptr(0) + i * 32);
#pragma unroll
            for (int s = 0; s < C::STEPS; s++) {
                if (s + 1 < C::STEPS) {
#pragma unroll
                    for (int i = 0; i < 4; i++) ar[(s + 1) & 1][i] = *reinterpret_cast<const bf16x8 *>(a_ptr(s + 1) + i * 32);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < C::TPW; j++)
                        acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j][s * 4 + i], ar[s & 1][i], acc[mt][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) park(other, mt);
        }
        __syncthreads();                                         // every wave is done reading the halo in `in`
        // epilogue: the product is taken transposed, C[row = output channel][col = pixel], so a lane holds runs of four
        // consecutive channels of ONE pixel (row = (reg & 3) + 8 (reg >> 2) + 4 kgrp): bias, ReLU, four bf16 = one 8-byte LDS
        // store per run into the output tile [128 pixels][COUT channels] (rows of OUT_STRIDE bytes) in the same buffer
        unsigned char *ot = in;
#pragma unroll
        for (int mt = 0; mt < C::MT; mt++) {
#pragma unroll
            for (int j = 0; j < C::TPW; j++) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int c0 = (ng * C::TPW + j) * 32 + q * 8 + kgrp * 4;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        v[e] = acc[mt][j][q * 4 + e] + bias_r[j][q][e];
                        if (a.relu) v[e] = fmaxf(v[e], 0.0f);
                    }
                    *reinterpret_cast<uint2 *>(ot + ((pg * C::MT + mt) * 32 + m) * C::OUT_STRIDE + c0 * 2) =
                        make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
                }
            }
        }
        __syncthreads();
        {
            int b, y0, x0;
            tile_origin(t, b, y0, x0);
#pragma unroll
            for (int u = 0; u < C::OVT; u++) {                   // 128 pixels x OVP vectors of 16 bytes
                const int v = tid + u * 256, px = v / C::OVP, cg = v % C::OVP;
                const int yy = y0 + (px >> 4), xx = x0 + (px & 15);
                if (yy < a.H && xx < a.W)
                    *reinterpret_cast<uint4 *>(a.y + (((size_t)b * a.H + yy) * a.W + xx) * COUT + cg * 8) =
                        *reinterpret_cast<const uint4 *>(ot + px * C::OUT_STRIDE + cg * 16);
            }
        }
        __syncthreads();                                         // this buffer is free for the tile after next
        cur ^= 1;
    }
}

template <int CIN, int COUT>
int launch_variant(const ConvArgs &a, int n_cus, hipStream_t stream) {
    using C = Cfg<CIN, COUT>;
    static LdsGrant grant;
    const size_t lds = 2 * (size_t)C::BUF;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv3x3_direct_kernel<CIN, COUT>), lds, grant)) return rc;
    const int grid = a.ntiles < n_cus ? a.ntiles : n_cus;        // persistent: one workgroup per CU
    hipLaunchKernelGGL((conv3x3_direct_kernel<CIN, COUT>), dim3(grid), dim3(256), lds, stream, a);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}
}  // namespace

bool conv3x3_direct_supported(int cin, int cout) {
    return (cin == 64 || cin == 128) && (cout == 64 || cout == 128);
}

int launch_conv3x3_direct(const void *x, const void *w, const float *bias, void *y, int B, int H, int W, int cin, int cout,
                          int relu, hipStream_t stream) {
    if (!conv3x3_direct_supported(cin, cout))
        return set_error(DSRG_ERR_INVALID, "conv3x3_direct: %d -> %d channels is not one of 64/128 -> 64/128", cin, cout);
    ConvArgs a;
    a.x = static_cast<const uint16_t *>(x); a.w = static_cast<const uint16_t *>(w); a.bias = bias;
    a.y = static_cast<uint16_t *>(y); a.B = B; a.H = H; a.W = W; a.relu = relu ? 1 : 0;
    a.tiles_x = (W + kTW - 1) / kTW; a.tiles_y = (H + kTH - 1) / kTH;
    const long nt = (long)B * a.tiles_x * a.tiles_y;
    if (B < 1 || H < 1 || W < 1 || nt > 0x7fffffffL) return set_error(DSRG_ERR_INVALID, "conv3x3_direct: bad shape");
    a.ntiles = (int)nt;
    static const int n_cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    if (cin == 64) return cout == 64 ? launch_variant<64, 64>(a, n_cus, stream) : launch_variant<64, 128>(a, n_cus, stream);
    return cout == 64 ? launch_variant<128, 64>(a, n_cus, stream) : launch_variant<128, 128>(a, n_cus, stream);
}

}  // namespace dsrg
