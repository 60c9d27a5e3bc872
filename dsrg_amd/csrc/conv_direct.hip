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
// Shape of the kernel: persistent workgroups of 4 waves, each wave with its share of the weight tensor in registers as MFMA
// fragments — 288 VGPRs at one wave per SIMD (the file has 512 at that occupancy), or 144 at two:
//     64 -> 64    a wave holds 32 of the outputs (1 tile x 36 k-steps = 144 VGPRs), 2 x 2 waves split outputs x pixels, and TWO
//                 workgroups share a CU: the 24 KB halo fetch of a workgroup is its only memory parallelism and the kernel is
//                 bound by exactly that (DSRG_CONV_OCC=1: all 64 outputs per wave, one workgroup per CU, 202 instead of 169 us)
//     64 -> 128   a wave holds 64 of the outputs, 2 x 2 waves split outputs x pixels (2 M-tiles of 32 pixels per wave)
//    128 -> 128   a wave holds 32 of the outputs (1 tile x 72 k-steps) and visits all 4 M-tiles
//    128 -> 64    a wave holds 32 of the outputs, 2 x 2 waves split outputs x pixels
// so LDS carries only the input: an 8 x 16 pixel output tile with its 10 x 18 halo (pixel stride = channels * 2 + 16 bytes:
// conflict-free 16-byte reads), double buffered — the next tile's halo is fetched into registers before the MFMAs
// (v_mfma_f32_32x32x16_bf16) of an M-tile and parked in the other buffer after them.  The output tile goes back through LDS
// for 16-byte coalesced NHWC stores.
#include "common.h"
#include <cstdlib>
#include <cstring>

#if defined(DSRG_EXP) && (DSRG_EXP & 16)
#define DSRG_DIRECT_DMA 0
#endif
#ifndef DSRG_DIRECT_DMA
#define DSRG_DIRECT_DMA 1                    // 1: the halo tiles of conv3x3_direct_kernel go global -> LDS by DMA (chunk-major image); 0: through registers (rounds 2-5)
#endif
#ifndef DSRG_EXP
#define DSRG_EXP 0                           // experiment builds (Makefile, EXP= / EXPSRC=conv_direct): 1 no halo fetch beyond the first
#endif                                       // tile, 2 no output stores, 4 no MFMAs — what binds the forward kernel (tools only)

namespace dsrg {
namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void;
constexpr int kTH = 8, kTW = 16;             // output tile (pixels): 4 M-tiles of (2 rows x 16 columns)
constexpr int kHH = kTH + 2, kHW = kTW + 2;  // halo tile

template <int CIN, int COUT, int TPW_ = (CIN == 64 ? 2 : 1)> struct Cfg {
    static constexpr int CG = CIN / 16;                    // 16-channel k-steps per tap
    static constexpr int KS = 9 * CG;                      // k-steps
    static constexpr int TPW = TPW_;                       // 32-channel output tiles per wave: TPW * KS * 4 = 288 (or 144) VGPRs
    static constexpr int WGS = TPW * KS * 4 <= 144 ? 2 : 1;   // workgroups per CU the register budget is cut for
    static constexpr int NG = COUT / (32 * TPW);           // waves across the output channels
    static constexpr int PG = 4 / NG;                      // waves across the pixels
    static constexpr int MT = 4 / PG;                      // 32-pixel M-tiles per wave
#if DSRG_EXP & 8
    static constexpr int KPS = 4;
#else
    static constexpr int KPS = WGS == 2 ? 2 : 4;           // k-steps per step of the MFMA loop (the operand prefetch unit)
#endif
    static constexpr int SPT = CG / KPS;                   // steps per tap
    static constexpr int STEPS = 9 * SPT;
    // halo by LDS-DMA (buffer_load ... lds) where it pays (measured at batch 16, round 6: 128 -> 128 152 -> 141 us; 64 -> 64 and
    // 128 -> 64 unchanged; 64 -> 128 slower, its M-tile pair spills): image CHUNK-MAJOR, [16-byte channel chunk c][halo pixel p],
    // 16 bytes each.  A DMA wave instruction lands 64 consecutive pixels of one chunk (lane-linear, as the instruction requires); a
    // fragment read's 16-lane group touches 16 (nearly) consecutive pixels of one chunk = consecutive 16-byte slots: conflict-free
    // without padding or swizzle, and the address is lane part (pixel * 16 + kgrp * PLANE) + compile-time constant (chunk * PLANE +
    // tap offset * 16): immediate offsets, as the padded pixel-major image has (an XOR-swizzled pixel-major image makes hipcc form
    // every step's address up front and spill the register-resident weights)
    static constexpr bool DMA = DSRG_DIRECT_DMA && CIN == 128 && COUT == 128;
    static constexpr int CPR = CIN / 8;                    // 16-byte chunks per halo pixel (8 / 16)
    static constexpr int PB = (kHH * kHW + 63) / 64;       // 64-pixel blocks per chunk plane (3)
    static constexpr int PLANE = PB * 1024;                // bytes per chunk plane
    static constexpr int NI = CPR * PB;                    // DMA wave instructions per halo tile (24 / 48)
    static constexpr int IN_STRIDE = CIN * 2 + 16;         // bytes per halo pixel in LDS (pixel-major image, through registers)
    static constexpr int OUT_STRIDE = COUT * 2 + 16;       // bytes per pixel of the output tile in LDS (2-way on the 8-byte stores)
    static constexpr int IN_BYTES = DMA && CPR * PLANE > kHH * kHW * IN_STRIDE ? CPR * PLANE : kHH * kHW * IN_STRIDE, OUT_BYTES = kTH * kTW * OUT_STRIDE;
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
    const uint16_t *mask;   // (B, H, W, COUT) bf16 or nullptr: outputs kept where mask > 0, else zeroed — the ReLU backward of the
                            // layer below when this launch is a data gradient (conv3x3_direct_kernel only)
    float *colsum;          // (workgroups, COUT) f32 or nullptr: every workgroup's column sums of what it stored (that layer's
                            // partial bias gradient)
};

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {   // v_cvt_pk_bf16_f32: round to nearest even
    f32x2 v = {lo, hi};
    bf16x2 b = __builtin_convertvector(v, bf16x2);
    return *reinterpret_cast<uint32_t *>(&b);
}

template <int CIN, int COUT, int TPW_, bool BWD = false>      // BWD: the masked data-gradient form (a.mask / a.colsum)
__global__ __launch_bounds__(256, (TPW_ * CIN <= 64 ? 2 : 1)) void conv3x3_direct_kernel(ConvArgs a) {
    using C = Cfg<CIN, COUT, TPW_>;
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

    // the bias of the output channels this lane writes: in registers where there is room, re-read in the epilogue otherwise
    constexpr bool kBiasRegs = C::WGS == 1;
    float bias_r[kBiasRegs ? C::TPW : 1][4][4];
    if (kBiasRegs) {
#pragma unroll
        for (int j = 0; j < C::TPW; j++)
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int e = 0; e < 4; e++)
                    bias_r[j][q][e] = a.bias ? a.bias[(ng * C::TPW + j) * 32 + q * 8 + kgrp * 4 + e] : 0.0f;
    }

    auto tile_origin = [&](int t, int &b, int &y0, int &x0) {
        const int per = a.tiles_x * a.tiles_y;
        b = t / per;
        const int r = t - b * per;
        y0 = (r / a.tiles_x) * kTH;
        x0 = (r % a.tiles_x) * kTW;
    };
    constexpr bool kDma = C::DMA && !BWD;                        // (the masked data-gradient instantiation spills 428 bytes a lane with it)
    // (kDma) the halo of tile t straight into LDS.  Wave instruction q = chunk * PB + block lands the 64 pixels [64 block, 64 block + 64) of
    // chunk plane `chunk`; lane = pixel inside the block; a pixel outside the image or beyond the halo reads zeros through the
    // descriptor's range check.  No staging registers, no ds_write pass; the wait moves to the barrier at the end of the tile.
    const rsrc_t rxh = make_rsrc(a.x, (size_t)a.B * a.H * a.W * CIN * 2);
    auto issue_halo = [&](int t, unsigned char *buf) {
        int b, y0, x0;
        tile_origin(t, b, y0, x0);
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        uint32_t poff[C::PB];                                    // this lane's pixel of each block: byte offset of its row in x, or out of range
#pragma unroll
        for (int blk = 0; blk < C::PB; blk++) {
            const int px = blk * 64 + lane, hy = px / kHW, hx = px - hy * kHW, yy = y0 - 1 + hy, xx = x0 - 1 + hx;
            const bool in = px < kHH * kHW && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
            poff[blk] = in ? (uint32_t)((b * a.H + yy) * a.W + xx) * (uint32_t)(CIN * 2) : 0x80000000u;
        }
#pragma unroll
        for (int u = 0; u < (C::NI + 3) / 4; u++) {
            const int q = wv + 4 * u;                            // (wave-uniform)
            if (q < C::NI) {
                const int c = q / C::PB, blk = q - c * C::PB;
                uint32_t off = poff[0];
#pragma unroll
                for (int k = 1; k < C::PB; k++) off = blk == k ? poff[k] : off;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rxh, (lds_void *)(buf + q * 1024), 16, off, (uint32_t)(c * 16), 0, 0);
            }
        }
    };
    // (!kDma) halo vector v of a tile: pixel (hy, hx) of the 10 x 18 halo, 16-byte channel group cg; chunk c of MT per thread
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
    if constexpr (kDma) {
        issue_halo(t, conv_lds);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll
        for (int c = 0; c < C::MT; c++) {
            fetch(t, c);
            park(conv_lds, c);
        }
    }
    __syncthreads();
    // masked launch: the column sums of this thread's channel group (tid % OVP) over all its tiles live in LDS behind the two
    // tile buffers (own slot per thread: no barrier) — the kernel has no registers to spare
    float *csl = reinterpret_cast<float *>(conv_lds + 2 * C::BUF) + tid * 8;
    if (BWD) {
#pragma unroll
        for (int e = 0; e < 8; e++) csl[e] = 0.0f;
    }
    // ... and the tile's mask rows are brought into LDS by DMA while the tile is computed (vector v = tid + 256 u of the store
    // phase at byte 16 v: a wave's 64 lanes land back to back, pixels outside the image read zeros through the descriptor's range)
    unsigned char *mtile = conv_lds + 2 * C::BUF + 256 * 8 * sizeof(float);
    const rsrc_t rmask = make_rsrc(a.mask, BWD ? (size_t)a.B * a.H * a.W * COUT * 2 : 0);
    for (; t < a.ntiles; t += gridDim.x) {
        const int tn = t + gridDim.x;
#if DSRG_EXP & 1
        const bool more = false;
#else
        const bool more = tn < a.ntiles;
#endif
        unsigned char *in = conv_lds + cur * C::BUF, *other = conv_lds + (cur ^ 1) * C::BUF;
        if (BWD) {
            int b, y0, x0;
            tile_origin(t, b, y0, x0);
            const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
            for (int u = 0; u < C::OVT; u++) {
                const int v = tid + u * 256, px = v / C::OVP, cg = v % C::OVP;
                const int yy = y0 + (px >> 4), xx = x0 + (px & 15);
                const uint32_t off = (yy < a.H && xx < a.W) ? (uint32_t)((b * a.H + yy) * a.W + xx) * (uint32_t)(COUT * 2) + (uint32_t)(cg * 16) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rmask, (lds_void *)(mtile + (u * 256 + wv * 64) * 16), 16, off, 0, 0, 0);
            }
        }
        if constexpr (kDma)
            if (more) issue_halo(tn, other);                     // the next tile's halo on its way while this one is computed
        f32x16 acc[C::MT][C::TPW];
#pragma unroll
        for (int mt = 0; mt < C::MT; mt++) {
            if constexpr (!kDma)
                if (more) fetch(tn, mt);                         // a slice of the next halo on its way while this M-tile computes
#pragma unroll
            for (int j = 0; j < C::TPW; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[mt][j][r] = 0.0f;
            // this wave's 32 pixels of the M-tile: tile rows 2 i, 2 i + 1; A[m][k]: m = pixel, k = (tap, channel)
            const int ty = 2 * (pg * C::MT + mt) + (m >> 4), tx = m & 15;
            // one wave per SIMD: nothing else hides the LDS latency of the pixel operand, so the four 16-byte reads of step
            // s + 1 are issued before the MFMAs of step s; sched_barrier pins that order (left alone, the scheduler sinks
            // every read next to its use and the wave waits out the LDS latency at every step)
            // fragment i of step s — chunk-major image: chunk h * 2 KPS + 2 i + kgrp of halo pixel (ty + tap / 3, tx + tap % 3);
            // pixel-major image: bytes [h * 32 KPS + 32 i + 16 kgrp, + 16) of that pixel's row
            const unsigned char *lane_base = kDma ? in + (ty * kHW + tx) * 16 + kgrp * C::PLANE : in + (ty * kHW + tx) * C::IN_STRIDE + kgrp * 16;
            auto a_ptr = [&](int s) {
                const int tap = s / C::SPT, h = s % C::SPT;
                return lane_base + ((tap / 3) * kHW + tap % 3) * (kDma ? 16 : C::IN_STRIDE) + h * (kDma ? 2 * C::KPS * C::PLANE : 32 * C::KPS);
            };
            constexpr int kFragStep = kDma ? 2 * C::PLANE : 32;  // fragment i + 1: two chunks further
            bf16x8 ar[2][C::KPS];
#pragma unroll
            for (int i = 0; i < C::KPS; i++) ar[0][i] = *reinterpret_cast<const bf16x8 *>(a_ptr(0) + i * kFragStep);
#pragma unroll
            for (int s = 0; s < C::STEPS; s++) {
                if (s + 1 < C::STEPS) {
#pragma unroll
                    for (int i = 0; i < C::KPS; i++) ar[(s + 1) & 1][i] = *reinterpret_cast<const bf16x8 *>(a_ptr(s + 1) + i * kFragStep);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < C::KPS; i++)
#pragma unroll
                    for (int j = 0; j < C::TPW; j++) {
#if DSRG_EXP & 4
                        if (s == 0) acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j][s * C::KPS + i], ar[s & 1][i], acc[mt][j], 0, 0, 0);
                        else asm volatile("" ::"v"(ar[s & 1][i]), "v"(wf[j][s * C::KPS + i]));
#else
                        acc[mt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j][s * C::KPS + i], ar[s & 1][i], acc[mt][j], 0, 0, 0);
#endif
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!kDma)
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
                    float v[4], bq[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (!kBiasRegs && a.bias) {
                        const float4 b4 = *reinterpret_cast<const float4 *>(a.bias + c0);
                        bq[0] = b4.x; bq[1] = b4.y; bq[2] = b4.z; bq[3] = b4.w;
                    }
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        v[e] = acc[mt][j][q * 4 + e] + (kBiasRegs ? bias_r[j][q][e] : bq[e]);
                        if (a.relu) v[e] = fmaxf(v[e], 0.0f);
                    }
                    *reinterpret_cast<uint2 *>(ot + ((pg * C::MT + mt) * 32 + m) * C::OUT_STRIDE + c0 * 2) =
                        make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
                }
            }
        }
        if (BWD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's mask rows have landed (it reads back its own)
        __syncthreads();
        {
            int b, y0, x0;
            tile_origin(t, b, y0, x0);
            float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < C::OVT; u++) {                   // 128 pixels x OVP vectors of 16 bytes
                const int v = tid + u * 256, px = v / C::OVP, cg = v % C::OVP;
                const int yy = y0 + (px >> 4), xx = x0 + (px & 15);
                uint4 val = *reinterpret_cast<const uint4 *>(ot + px * C::OUT_STRIDE + cg * 16);
                if (BWD) {
                    uint32_t w4[4] = {val.x, val.y, val.z, val.w};
                    const uint4 mkv = *reinterpret_cast<const uint4 *>(mtile + v * 16);
                    const uint32_t y4[4] = {mkv.x, mkv.y, mkv.z, mkv.w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        w4[e] &= ((int16_t)(y4[e] & 0xffffu) > 0 ? 0x0000ffffu : 0u) | ((int32_t)y4[e] >= 0x10000 ? 0xffff0000u : 0u);
                        cs[2 * e] += __uint_as_float(w4[e] << 16);          // (pixels outside the image: mask 0)
                        cs[2 * e + 1] += __uint_as_float(w4[e] & 0xffff0000u);
                    }
                    val = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                }
#if DSRG_EXP & 2
                if (yy < a.H && xx < a.W && val.x == 0x12345678u)
#else
                if (yy < a.H && xx < a.W)
#endif
                    *reinterpret_cast<uint4 *>(a.y + (((size_t)b * a.H + yy) * a.W + xx) * COUT + cg * 8) = val;
            }
            if (BWD) {
#pragma unroll
                for (int e = 0; e < 8; e++) csl[e] += cs[e];
            }
        }
        if constexpr (kDma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next halo have landed (the barrier: everybody's)
        __syncthreads();                                         // this buffer is free for the tile after next
#if !(DSRG_EXP & 1)
        cur ^= 1;
#endif
    }
    if (BWD) {                                                   // one partial bias row per workgroup, fixed order
        const float *red = reinterpret_cast<const float *>(conv_lds + 2 * C::BUF);        // [256 threads][8]
        __syncthreads();
        if (tid < COUT) {
            const int cg = tid >> 3, e = tid & 7;
            float s0 = 0.0f, s1 = 0.0f;
#pragma unroll 4
            for (int i = 0; i < 256 / C::OVP; i += 2) {
                s0 += red[(cg + C::OVP * i) * 8 + e];
                s1 += red[(cg + C::OVP * (i + 1)) * 8 + e];
            }
            a.colsum[(size_t)blockIdx.x * COUT + tid] = s0 + s1;
        }
    }
}

template <int CIN, int COUT, int TPW_>
int variant_grid(const ConvArgs &a, int n_cus) {
    const int slots = n_cus * Cfg<CIN, COUT, TPW_>::WGS;
    return a.ntiles < slots ? a.ntiles : slots;
}
template <int CIN, int COUT, int TPW_>
int launch_variant(const ConvArgs &a, int n_cus, hipStream_t stream) {
    using C = Cfg<CIN, COUT, TPW_>;
    static LdsGrant grant, grant_b;
    // (the DMA halo marks a lane outside the image with byte offset 2^31, which must lie beyond the descriptor's range)
    if (C::DMA && !a.mask && (size_t)a.B * a.H * a.W * CIN * 2 >= ((size_t)1 << 31)) return DSRG_ERR_UNSUPPORTED;
    const int grid = variant_grid<CIN, COUT, TPW_>(a, n_cus);    // persistent: one or two workgroups per CU
    if (a.mask) {                                                // masked data gradient: + 8 KB of column sums
        const size_t lds = 2 * (size_t)C::BUF + 256 * 8 * sizeof(float) + (size_t)kTH * kTW * COUT * 2;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv3x3_direct_kernel<CIN, COUT, TPW_, true>), lds, grant_b)) return rc;
        hipLaunchKernelGGL((conv3x3_direct_kernel<CIN, COUT, TPW_, true>), dim3(grid), dim3(256), lds, stream, a);
    } else {
        const size_t lds = 2 * (size_t)C::BUF;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv3x3_direct_kernel<CIN, COUT, TPW_>), lds, grant)) return rc;
        hipLaunchKernelGGL((conv3x3_direct_kernel<CIN, COUT, TPW_>), dim3(grid), dim3(256), lds, stream, a);
    }
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}
// ------------------------------------------------------------------------------------------------------------------
// The first layer, 3 -> 64 channels at full resolution (conv1_1, train-s.prototxt:44-64): output-bandwidth-bound (211 MB at
// batch 16), which MIOpen's forward (96 us) + a separate bias add (80 us) + ReLU (63 us) passes over three times.  Same tile
// and epilogue as above; the halo keeps 4 channels per pixel (3 + a zero: 8 bytes), k = tap * 4 + channel padded from 36 to
// 48 = three 16-wide k-steps, so a lane's fragment is two 8-byte LDS reads (two taps) and the taps past the ninth meet zero
// weights.
constexpr int kC3Pix = 8;                                  // bytes per halo pixel

__global__ __launch_bounds__(256) void conv3x3_c3_kernel(ConvArgs a) {   // x (B,H,W,3), w (64,3,3,3) = [o][ky][kx][c], y (B,H,W,64)
    __shared__ __attribute__((aligned(16))) unsigned char halo[2][kHH * kHW * kC3Pix];
    __shared__ __attribute__((aligned(16))) unsigned char ot[kTH * kTW * 144];
    constexpr int kOutStride = 144;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, kgrp = lane >> 5;

    bf16x8 wf[2][3];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int s3 = 0; s3 < 3; s3++) {
            uint16_t v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int tap = 4 * s3 + 2 * kgrp + (e >> 2), c = e & 3;
                v[e] = (tap < 9 && c < 3) ? a.w[((nt * 32 + m) * 9 + tap) * 3 + c] : (uint16_t)0;
            }
            uint4 pk = make_uint4(v[0] | (uint32_t)v[1] << 16, v[2] | (uint32_t)v[3] << 16, v[4] | (uint32_t)v[5] << 16,
                                  v[6] | (uint32_t)v[7] << 16);
            wf[nt][s3] = *reinterpret_cast<bf16x8 *>(&pk);
        }
    float bias_r[2][4][4];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int e = 0; e < 4; e++) bias_r[nt][q][e] = a.bias ? a.bias[nt * 32 + q * 8 + kgrp * 4 + e] : 0.0f;
    // byte offsets of this lane's two taps per k-step inside the halo (taps past the ninth: any pixel, their weights are zero)
    int toff[3][2];
#pragma unroll
    for (int s3 = 0; s3 < 3; s3++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int tap = min(4 * s3 + 2 * kgrp + h, 8);
            toff[s3][h] = ((tap / 3) * kHW + tap % 3) * kC3Pix;
        }

    auto tile_origin = [&](int t, int &b, int &y0, int &x0) {
        const int per = a.tiles_x * a.tiles_y;
        b = t / per;
        const int r = t - b * per;
        y0 = (r / a.tiles_x) * kTH;
        x0 = (r % a.tiles_x) * kTW;
    };
    uint2 pre = make_uint2(0u, 0u);
    auto fetch = [&](int t) {                                    // threads 0 .. 179: one halo pixel each
        int b, y0, x0;
        tile_origin(t, b, y0, x0);
        pre = make_uint2(0u, 0u);
        if (tid < kHH * kHW) {
            const int hy = tid / kHW, hx = tid - hy * kHW, yy = y0 - 1 + hy, xx = x0 - 1 + hx;
            if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
                const uint16_t *p = a.x + (((size_t)b * a.H + yy) * a.W + xx) * 3;
                pre = make_uint2(p[0] | (uint32_t)p[1] << 16, p[2]);
            }
        }
    };
    auto park = [&](unsigned char *buf) {
        if (tid < kHH * kHW) *reinterpret_cast<uint2 *>(buf + tid * kC3Pix) = pre;
    };

    int t = blockIdx.x, cur = 0;
    if (t >= a.ntiles) return;
    fetch(t);
    park(halo[0]);
    __syncthreads();
    const int ty = 2 * wave + (m >> 4), tx = m & 15;
    const int pix_off = (ty * kHW + tx) * kC3Pix;
    for (; t < a.ntiles; t += gridDim.x) {
        const int tn = t + gridDim.x;
        const bool more = tn < a.ntiles;
        if (more) fetch(tn);
        const unsigned char *in = halo[cur] + pix_off;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; r++) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
#pragma unroll
        for (int s3 = 0; s3 < 3; s3++) {
            const uint2 lo = *reinterpret_cast<const uint2 *>(in + toff[s3][0]), hi = *reinterpret_cast<const uint2 *>(in + toff[s3][1]);
            uint4 pk = make_uint4(lo.x, lo.y, hi.x, hi.y);
            const bf16x8 ar = *reinterpret_cast<bf16x8 *>(&pk);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][s3], ar, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][s3], ar, acc1, 0, 0, 0);
        }
        // transposed product as above: a lane holds runs of four consecutive output channels of one pixel
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int c0 = nt * 32 + q * 8 + kgrp * 4;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    v[e] = (nt ? acc1[q * 4 + e] : acc0[q * 4 + e]) + bias_r[nt][q][e];
                    if (a.relu) v[e] = fmaxf(v[e], 0.0f);
                }
                *reinterpret_cast<uint2 *>(ot + (wave * 32 + m) * kOutStride + c0 * 2) = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
            }
        }
        if (more) park(halo[cur ^ 1]);
        __syncthreads();                                         // output tile and next halo complete
        {
            int b, y0, x0;
            tile_origin(t, b, y0, x0);
#pragma unroll
            for (int u = 0; u < 4; u++) {                        // 128 pixels x 8 vectors of 16 bytes
                const int v = tid + u * 256, px = v >> 3, cg = v & 7;
                const int yy = y0 + (px >> 4), xx = x0 + (px & 15);
                if (yy < a.H && xx < a.W)
                    *reinterpret_cast<uint4 *>(a.y + (((size_t)b * a.H + yy) * a.W + xx) * 64 + cg * 8) =
                        *reinterpret_cast<const uint4 *>(ot + px * kOutStride + cg * 16);
            }
        }
        __syncthreads();                                         // the output tile is free again
        cur ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same layers: gw[o][tap][c] = sum over pixels of g[px][o] * x[px + tap][c], a GEMM whose reduction
// runs over the pixels — the slow axis of both NHWC operands.  gfx950's transposing LDS read (ds_read_b64_tr_b16: a
// 16-lane group reads a [4 pixels][16 channels] block, lane i supplying the 8-byte address of row i / 4, chunk i % 4, and
// lane c receiving column c) turns the plain NHWC tiles in LDS into MFMA fragments with 8 consecutive pixels per lane, so
// the tiles are staged exactly as in the forward kernel (a 10 x 18 halo of x, the 8 x 16 tile of g) and the nine taps are
// nine pixel offsets into the same x tile.  Pixel strides of 64 bytes (mod 256) keep the four rows of a read on distinct
// bank quarters.
//
// Work split: the reduction is split over persistent workgroups, each accumulating its partial gw for a 64-channel slice of
// x in registers across all its tiles: a wave owns 32 channels of g x 32 channels of x x 9 taps = 144 accumulator VGPRs, a
// workgroup has one wave per pair (4 waves for 64-channel g, 8 for 128: two waves per SIMD); 128-channel x is two slices
// handled by alternating workgroups.
// The partials (one per workgroup) are summed in a fixed order by a second kernel that also rounds to bf16: deterministic.
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4v;
#define DSRG_LDS __attribute__((address_space(3)))

template <int CO_TILES> struct WCfg {                      // CO_TILES = channels of g / 32 (2 or 4)
    static constexpr int COUT = 32 * CO_TILES;
    static constexpr int THREADS = CO_TILES * 128;         // one wave per (32 channels of g, 32 of the 64 channels of x)
    static constexpr int NT = 9;                           // accumulator tiles per wave: the taps
    static constexpr int SX = 192;                         // bytes per halo pixel of the x slice (128 + 64)
    static constexpr int SG = COUT * 2 + 64;               // bytes per pixel of g (192 / 320)
    static constexpr int X_BYTES = kHH * kHW * SX, G_BYTES = kTH * kTW * SG, BUF = X_BYTES + G_BYTES;
    static constexpr int XV = kHH * kHW * 8, XVT = (XV + THREADS - 1) / THREADS;   // 16-byte vectors of the x tile, per thread (6 / 3)
    static constexpr int GVP = COUT / 8, GVT = kTH * kTW * GVP / THREADS;          // of the g tile (4)
    static constexpr int PART = COUT * 9 * 64;             // floats of one workgroup's partial
    // 64-channel g: 4 waves of < 256 VGPRs, so two workgroups share a CU (each with ONE tile buffer: 59 KB) and one's loads and
    // barriers hide behind the other's MFMAs; 128-channel g: 8 waves, one workgroup per CU, two buffers
    static constexpr int WGS = CO_TILES == 2 ? 2 : 1, NBUF = WGS == 2 ? 1 : 2;
};

struct WgradArgs {
    const uint16_t *x;      // (B, H, W, cin) bf16
    const uint16_t *g;      // (B, H, W, cout) bf16
    float *part;            // (workgroups, cout, 9, 64) f32
    int B, H, W, cin, tiles_x, tiles_y, nitems, roles;
};

__device__ __forceinline__ bf16x8 tr_frag(DSRG_LDS unsigned char *p, int stride4) {
    bf16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(reinterpret_cast<DSRG_LDS bf16x4v *>(p));
    bf16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(reinterpret_cast<DSRG_LDS bf16x4v *>(p + stride4));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int CO_TILES>
__global__ __launch_bounds__(CO_TILES * 128, (CO_TILES == 2 ? 2 : 1)) void conv3x3_wgrad_kernel(WgradArgs a) {
    using C = WCfg<CO_TILES>;
    extern __shared__ __attribute__((aligned(16))) unsigned char conv_lds[];
    DSRG_LDS unsigned char *lds = (DSRG_LDS unsigned char *)conv_lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, gq = lane >> 4;
    const int otile = wave % CO_TILES;                             // this wave's 32 channels of g
    const int ct0 = wave / CO_TILES;                               // its 32-channel tile of the x slice
    const int role = blockIdx.x % a.roles;                         // the 64-channel slice of x this workgroup reduces
    // the address this lane supplies to a transposing read: pixel (gq / 2) * 8 + i16 / 4 of the k-step (the second read of a
    // fragment 4 pixels further), channels (gq % 2) * 16 + 4 (i16 % 4) .. + 3 of the 32-channel tile
    const int rowsel = (gq >> 1) * 8 + (i16 >> 2), colsel = (gq & 1) * 16 + 4 * (i16 & 3);
    const int x_lane = rowsel * C::SX + (ct0 * 32 + colsel) * 2;
    const int g_lane = C::X_BYTES + rowsel * C::SG + (otile * 32 + colsel) * 2;

    auto tile_origin = [&](int item, int &b, int &y0, int &x0) {
        const int t = item / a.roles, per = a.tiles_x * a.tiles_y;
        b = t / per;
        const int r = t - b * per;
        y0 = (r / a.tiles_x) * kTH;
        x0 = (r % a.tiles_x) * kTW;
    };
    uint4 prx[C::XVT], prg[C::GVT];
    auto fetch = [&](int item) {
        int b, y0, x0;
        tile_origin(item, b, y0, x0);
#pragma unroll
        for (int u = 0; u < C::XVT; u++) {
            const int v = tid + u * C::THREADS;
            uint4 val = make_uint4(0u, 0u, 0u, 0u);
            if (v < C::XV) {
                const int px = v >> 3, cg = v & 7, hy = px / kHW, hx = px - hy * kHW;
                const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
                if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W)
                    val = *reinterpret_cast<const uint4 *>(a.x + (((size_t)b * a.H + yy) * a.W + xx) * a.cin + role * 64 + cg * 8);
            }
            prx[u] = val;
        }
#pragma unroll
        for (int u = 0; u < C::GVT; u++) {
            const int v = tid + u * C::THREADS, px = v / C::GVP, cg = v % C::GVP;
            const int yy = y0 + (px >> 4), xx = x0 + (px & 15);
            uint4 val = make_uint4(0u, 0u, 0u, 0u);                 // pixels past the image edge add nothing
            if (yy < a.H && xx < a.W)
                val = *reinterpret_cast<const uint4 *>(a.g + (((size_t)b * a.H + yy) * a.W + xx) * C::COUT + cg * 8);
            prg[u] = val;
        }
    };
    auto park = [&](unsigned char *buf) {
#pragma unroll
        for (int u = 0; u < C::XVT; u++) {
            const int v = tid + u * C::THREADS;
            if (v < C::XV) *reinterpret_cast<uint4 *>(buf + (v >> 3) * C::SX + (v & 7) * 16) = prx[u];
        }
#pragma unroll
        for (int u = 0; u < C::GVT; u++) {
            const int v = tid + u * C::THREADS;
            *reinterpret_cast<uint4 *>(buf + C::X_BYTES + (v / C::GVP) * C::SG + (v % C::GVP) * 16) = prg[u];
        }
    };

    f32x16 acc[C::NT];
#pragma unroll
    for (int n = 0; n < C::NT; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[n][r] = 0.0f;

    int item = blockIdx.x, cur = 0;
    if (item < a.nitems) {
        fetch(item);
        park(conv_lds);
    }
    __syncthreads();
    for (; item < a.nitems; item += gridDim.x) {
        const int nxt = item + gridDim.x;
        const bool more = nxt < a.nitems;
        if (more) fetch(nxt);                                    // the next tiles on their way while this one is reduced
        DSRG_LDS unsigned char *in = lds + (C::NBUF == 2 ? cur : 0) * C::BUF;
        DSRG_LDS unsigned char *xb = in + x_lane, *gb = in + g_lane;
        // k-step ks = tile row ks of g (16 pixels); accumulator tile = tap (dy, dx): its x fragment starts at halo pixel
        // (ks + dy, dx).  So the loop runs over the ten halo rows R: the three fragments of row R (dx = 0, 1, 2) are read
        // once and meet the g fragments of rows R, R - 1, R - 2 (dy = 0, 1, 2) — 76 transposing reads per tile instead of 160.
        // The reads of row R + 1 are issued before the MFMAs of row R (see the forward kernel).
        bf16x8 ga[4], bx[2][3];
        ga[0] = tr_frag(gb, 4 * C::SG);
#pragma unroll
        for (int u = 0; u < 3; u++) bx[0][u] = tr_frag(xb + u * C::SX, 4 * C::SX);
#pragma unroll
        for (int R = 0; R < kHH; R++) {
            if (R + 1 < kHH) {
#pragma unroll
                for (int u = 0; u < 3; u++) bx[(R + 1) & 1][u] = tr_frag(xb + ((R + 1) * kHW + u) * C::SX, 4 * C::SX);
                if (R + 1 < kTH) ga[(R + 1) & 3] = tr_frag(gb + (R + 1) * kTW * C::SG, 4 * C::SG);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dy = 0; dy < 3; dy++) {
                const int ks = R - dy;
                if (ks >= 0 && ks < kTH) {
#pragma unroll
                    for (int dx = 0; dx < 3; dx++)
                        acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[ks & 3], bx[R & 1][dx], acc[dy * 3 + dx], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (C::NBUF == 2) {
            if (more) park(conv_lds + (cur ^ 1) * C::BUF);
            __syncthreads();                                     // the other buffer is complete, this one is free
            cur ^= 1;
        } else {
            __syncthreads();                                     // every wave is done with the tile
            if (more) park(conv_lds);
            __syncthreads();
        }
    }
    // C[row = channel of g][col = channel of x]: lane holds column lane % 32, rows (reg & 3) + 8 (reg >> 2) + 4 (lane / 32)
    float *pp = a.part + (size_t)blockIdx.x * C::PART;
#pragma unroll
    for (int n = 0; n < C::NT; n++) {
        const int tap = n, c = ct0 * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int o = otile * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            pp[(o * 9 + tap) * 64 + c] = acc[n][r];
        }
    }
}

// gw[o][tap][c] (bf16, the memory of a channels_last (cout, cin, 3, 3) tensor) = sum of the partials of slice c / 64 in
// workgroup order.  1024 threads = 64 consecutive channels x 16 interleaved sixteenths of the workgroups (loads of many
// partials in flight per thread: the sum is a latency chain otherwise), combined in a fixed order.
constexpr int kRedChunks = 16;
template <bool F32_OUT>
__global__ __launch_bounds__(1024) void conv3x3_wgrad_reduce_kernel(const float *part, void *gw, int nwg, int roles, int cout,
                                                                    int cin) {
    __shared__ float red[kRedChunks][64];
    const int el = threadIdx.x & 63, ch = threadIdx.x >> 6;
    const int e0 = blockIdx.x * 64;                              // first element (o, tap, c) of this block; c % 64 == 0
    const int o = e0 / (9 * cin), rem = e0 - o * 9 * cin, tap = rem / cin, c0 = rem - tap * cin, role = c0 >> 6;
    const size_t stride = (size_t)cout * 9 * 64;
    const float *p = part + ((size_t)o * 9 + tap) * 64 + el;
    float s = 0.0f;
    for (int w = role + ch * roles; w < nwg; w += kRedChunks * roles) s += p[(size_t)w * stride];
    red[ch][el] = s;
    __syncthreads();
    if (ch == 0) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < kRedChunks; k++) t += red[k][el];
        if (F32_OUT) {
            static_cast<float *>(gw)[(size_t)e0 + el] = t;
        } else {
            f32x2 v = {t, 0.0f};
            bf16x2 bv = __builtin_convertvector(v, bf16x2);
            static_cast<uint16_t *>(gw)[(size_t)e0 + el] = *reinterpret_cast<uint16_t *>(&bv);
        }
    }
}

// Weight gradient of the first layer (x 3 channels, g 64): gw[o][tap][c] = sum over pixels of g[px][o] * x[px + tap][c].  The
// nine taps of a pixel are gathered from the 4-channel halo into an im2col tile [pixel][n = tap * 4 + c] (36 of 64 columns
// used, the rest stay zero), after which it is the same transposing-read GEMM as above with one accumulator tile per wave
// (32 channels of g x 32 columns).  Partials [workgroup][64][64] f32, summed by conv3x3_c3_wgrad_reduce_kernel.
constexpr int kC3Row = 192;                                // bytes per pixel row of the g tile and of the im2col tile (128 + 64)

__global__ __launch_bounds__(256) void conv3x3_c3_wgrad_kernel(WgradArgs a) {      // x (B,H,W,3), g (B,H,W,64)
    extern __shared__ __attribute__((aligned(16))) unsigned char conv_lds[];
    constexpr int kHaloB = kHH * kHW * kC3Pix, kTileB = kTH * kTW * kC3Row;      // 1 440, 24 576
    auto halo = [&](int b) { return conv_lds + b * kHaloB; };
    auto gt = [&](int b) { return conv_lds + 2 * kHaloB + b * kTileB; };
    unsigned char *xc = conv_lds + 2 * kHaloB + 2 * kTileB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, gq = lane >> 4;
    const int otile = wave & 1, ntile = wave >> 1;
    const int rowsel = (gq >> 1) * 8 + (i16 >> 2), colsel = (gq & 1) * 16 + 4 * (i16 & 3);
    const int g_lane = rowsel * kC3Row + (otile * 32 + colsel) * 2, x_lane = rowsel * kC3Row + (ntile * 32 + colsel) * 2;
    for (int v = tid; v < kTH * kTW * kC3Row / 16; v += 256) reinterpret_cast<uint4 *>(xc)[v] = make_uint4(0u, 0u, 0u, 0u);

    auto tile_origin = [&](int t, int &b, int &y0, int &x0) {
        const int per = a.tiles_x * a.tiles_y;
        b = t / per;
        const int r = t - b * per;
        y0 = (r / a.tiles_x) * kTH;
        x0 = (r % a.tiles_x) * kTW;
    };
    uint2 prx = make_uint2(0u, 0u);
    uint4 prg[4];
    auto fetch = [&](int t) {
        int b, y0, x0;
        tile_origin(t, b, y0, x0);
        prx = make_uint2(0u, 0u);
        if (tid < kHH * kHW) {
            const int hy = tid / kHW, hx = tid - hy * kHW, yy = y0 - 1 + hy, xx = x0 - 1 + hx;
            if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
                const uint16_t *p = a.x + (((size_t)b * a.H + yy) * a.W + xx) * 3;
                prx = make_uint2(p[0] | (uint32_t)p[1] << 16, p[2]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int v = tid + u * 256, px = v >> 3, cg = v & 7;
            const int yy = y0 + (px >> 4), xx = x0 + (px & 15);
            uint4 val = make_uint4(0u, 0u, 0u, 0u);                 // pixels past the image edge add nothing
            if (yy < a.H && xx < a.W) val = *reinterpret_cast<const uint4 *>(a.g + (((size_t)b * a.H + yy) * a.W + xx) * 64 + cg * 8);
            prg[u] = val;
        }
    };
    auto park = [&](int buf) {
        if (tid < kHH * kHW) *reinterpret_cast<uint2 *>(halo(buf) + tid * kC3Pix) = prx;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int v = tid + u * 256;
            *reinterpret_cast<uint4 *>(gt(buf) + (v >> 3) * kC3Row + (v & 7) * 16) = prg[u];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    int t = blockIdx.x, cur = 0;
    if (t < a.nitems) {
        fetch(t);
        park(0);
    }
    __syncthreads();
    const int cpx = tid >> 1, chalf = tid & 1;                   // im2col: two threads per pixel, taps 0-4 and 5-8
    const int cty = cpx >> 4, ctx_ = cpx & 15;
    for (; t < a.nitems; t += gridDim.x) {
        const int tn = t + gridDim.x;
        const bool more = tn < a.nitems;
        if (more) fetch(tn);
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int tap = chalf * 5 + k;
            if (tap < 9)
                *reinterpret_cast<uint2 *>(xc + cpx * kC3Row + tap * 8) =
                    *reinterpret_cast<const uint2 *>(halo(cur) + ((cty + tap / 3) * kHW + ctx_ + tap % 3) * kC3Pix);
        }
        __syncthreads();                                         // im2col tile complete
        DSRG_LDS unsigned char *gb = (DSRG_LDS unsigned char *)gt(cur) + g_lane, *xb = (DSRG_LDS unsigned char *)xc + x_lane;
#pragma unroll
        for (int ks = 0; ks < kTH; ks++) {
            const bf16x8 ga = tr_frag(gb + ks * kTW * kC3Row, 4 * kC3Row), bx = tr_frag(xb + ks * kTW * kC3Row, 4 * kC3Row);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga, bx, acc, 0, 0, 0);
        }
        if (more) park(cur ^ 1);
        __syncthreads();                                         // next tiles parked, im2col tile free
        cur ^= 1;
    }
    float *pp = a.part + (size_t)blockIdx.x * 64 * 64;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int o = otile * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        pp[o * 64 + ntile * 32 + (lane & 31)] = acc[r];
    }
}

// gw[o][tap][c] (bf16, (64, 3, 3, 3) channels_last) = sum over the workgroups of part[wg][o][tap * 4 + c], in a fixed order;
// as above: 64 consecutive entries of the 64 x 64 partial x 16 interleaved sixteenths of the workgroups per block
template <bool F32_OUT>
__global__ __launch_bounds__(1024) void conv3x3_c3_wgrad_reduce_kernel(const float *part, void *gw, int nwg) {
    __shared__ float red[kRedChunks][64];
    const int el = threadIdx.x & 63, ch = threadIdx.x >> 6;
    const int o = blockIdx.x;                                    // one row of the partial: n = tap * 4 + c
    const float *p = part + o * 64 + el;
    float s = 0.0f;
    for (int w = ch; w < nwg; w += kRedChunks) s += p[(size_t)w * 4096];
    red[ch][el] = s;
    __syncthreads();
    const int tap = el >> 2, c = el & 3;
    if (ch == 0 && tap < 9 && c < 3) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < kRedChunks; k++) t += red[k][el];
        if (F32_OUT) {
            static_cast<float *>(gw)[(o * 9 + tap) * 3 + c] = t;
        } else {
            f32x2 v = {t, 0.0f};
            bf16x2 bv = __builtin_convertvector(v, bf16x2);
            static_cast<uint16_t *>(gw)[(o * 9 + tap) * 3 + c] = *reinterpret_cast<uint16_t *>(&bv);
        }
    }
}

template <int CO_TILES>
int launch_wgrad_variant(const WgradArgs &a, int grid, hipStream_t stream) {
    using C = WCfg<CO_TILES>;
    static LdsGrant grant;
    const size_t lds = C::NBUF * (size_t)C::BUF;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv3x3_wgrad_kernel<CO_TILES>), lds, grant)) return rc;
    hipLaunchKernelGGL((conv3x3_wgrad_kernel<CO_TILES>), dim3(grid), dim3(C::THREADS), lds, stream, a);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

int device_cus() {
    static const int n_cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 2) n = 256;
        return n;
    }();
    return n_cus;
}

bool wgrad_supported(int cin, int cout) {
    return (cin == 64 && (cout == 64 || cout == 128)) || (cin == 128 && cout == 128) || (cin == 3 && cout == 64);
}

// workgroups of the reduction for (B, H, W): a multiple of the slices of x
int wgrad_grid(int B, int H, int W, int cin, int cout) {
    const int roles = cin < 64 ? 1 : cin / 64;
    const long items = (long)B * ((W + kTW - 1) / kTW) * ((H + kTH - 1) / kTH) * roles;
    const int per_cu = cout == 64 ? 2 : 1;                       // workgroups that share a CU (WCfg::WGS; the 3-channel kernel too)
    const long cap = (long)device_cus() * per_cu / roles * roles;
    return (int)(items < cap ? items : cap);
}
}  // namespace

size_t conv3x3_wgrad_workspace(int B, int H, int W, int cin, int cout) {
    if (!wgrad_supported(cin, cout) || B < 1 || H < 1 || W < 1) return 0;
    if (cin == 3) return (size_t)wgrad_grid(B, H, W, cin, cout) * 64 * 64 * sizeof(float);
    return (size_t)wgrad_grid(B, H, W, cin, cout) * cout * 9 * 64 * sizeof(float);
}

int launch_conv3x3_wgrad(const void *x, const void *g, void *gw, float *workspace, size_t workspace_bytes, int B, int H, int W,
                         int cin, int cout, hipStream_t stream, int out_f32) {
    if (!wgrad_supported(cin, cout))
        return set_error(DSRG_ERR_INVALID, "conv3x3_wgrad: %d -> %d channels is not one of 3 -> 64, 64 -> 64, 64 -> 128, 128 -> 128", cin, cout);
    WgradArgs a;
    a.x = static_cast<const uint16_t *>(x); a.g = static_cast<const uint16_t *>(g); a.part = workspace;
    a.B = B; a.H = H; a.W = W; a.cin = cin; a.roles = cin < 64 ? 1 : cin / 64;
    a.tiles_x = (W + kTW - 1) / kTW; a.tiles_y = (H + kTH - 1) / kTH;
    const long items = (long)B * a.tiles_x * a.tiles_y * a.roles;
    if (B < 1 || H < 1 || W < 1 || items > 0x7fffffffL) return set_error(DSRG_ERR_INVALID, "conv3x3_wgrad: bad shape");
    a.nitems = (int)items;
    const int grid = wgrad_grid(B, H, W, cin, cout);
    if (workspace_bytes < conv3x3_wgrad_workspace(B, H, W, cin, cout))
        return set_error(DSRG_ERR_INVALID, "conv3x3_wgrad: workspace of %zu bytes, %zu needed", workspace_bytes,
                         conv3x3_wgrad_workspace(B, H, W, cin, cout));
    if (cin == 3) {
        static LdsGrant grant;
        const size_t lds = 2 * kHH * kHW * kC3Pix + 3 * kTH * kTW * kC3Row;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv3x3_c3_wgrad_kernel), lds, grant)) return rc;
        hipLaunchKernelGGL(conv3x3_c3_wgrad_kernel, dim3(grid), dim3(256), lds, stream, a);
        DSRG_LAUNCH_CHECK();
        if (out_f32) hipLaunchKernelGGL(conv3x3_c3_wgrad_reduce_kernel<true>, dim3(64), dim3(1024), 0, stream, workspace, gw, grid);
        else hipLaunchKernelGGL(conv3x3_c3_wgrad_reduce_kernel<false>, dim3(64), dim3(1024), 0, stream, workspace, gw, grid);
        DSRG_LAUNCH_CHECK();
        return DSRG_OK;
    }
    if (int rc = cout == 64 ? launch_wgrad_variant<2>(a, grid, stream) : launch_wgrad_variant<4>(a, grid, stream)) return rc;
    if (out_f32)
        hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel<true>, dim3(cout * 9 * cin / 64), dim3(1024), 0, stream, workspace, gw, grid, a.roles,
                           cout, cin);
    else
        hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel<false>, dim3(cout * 9 * cin / 64), dim3(1024), 0, stream, workspace, gw, grid, a.roles,
                           cout, cin);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

bool conv3x3_direct_supported(int cin, int cout) {
    return ((cin == 64 || cin == 128) && (cout == 64 || cout == 128)) || (cin == 3 && cout == 64);
}

size_t conv3x3_direct_colsum_workspace(int cout) { return (size_t)2 * device_cus() * (size_t)cout * sizeof(float); }

int launch_conv3x3_direct(const void *x, const void *w, const float *bias, void *y, int B, int H, int W, int cin, int cout,
                          int relu, hipStream_t stream, const void *mask, float *colsum, void *colsum_ws, size_t colsum_ws_bytes) {
    if (!conv3x3_direct_supported(cin, cout))
        return set_error(DSRG_ERR_INVALID, "conv3x3_direct: %d -> %d channels is not one of 64/128 -> 64/128 or 3 -> 64", cin, cout);
    if ((mask || colsum) && (cin == 3 || !mask || !colsum))
        return set_error(DSRG_ERR_UNSUPPORTED, "conv3x3_direct: the masked form takes 64 / 128 channels, mask and bias gradient together");
    if (mask && (long long)B * H * W * cout * 2 >= 0x7fffffffLL)
        return set_error(DSRG_ERR_UNSUPPORTED, "conv3x3_direct: mask too large for a 32-bit buffer offset");
    if (colsum && (!colsum_ws || colsum_ws_bytes < conv3x3_direct_colsum_workspace(cout)))
        return set_error(DSRG_ERR_INVALID, "conv3x3_direct: column-sum scratch missing or too small");
    ConvArgs a;
    a.mask = static_cast<const uint16_t *>(mask); a.colsum = colsum ? static_cast<float *>(colsum_ws) : nullptr;
    a.x = static_cast<const uint16_t *>(x); a.w = static_cast<const uint16_t *>(w); a.bias = bias;
    a.y = static_cast<uint16_t *>(y); a.B = B; a.H = H; a.W = W; a.relu = relu ? 1 : 0;
    a.tiles_x = (W + kTW - 1) / kTW; a.tiles_y = (H + kTH - 1) / kTH;
    const long nt = (long)B * a.tiles_x * a.tiles_y;
    if (B < 1 || H < 1 || W < 1 || nt > 0x7fffffffL) return set_error(DSRG_ERR_INVALID, "conv3x3_direct: bad shape");
    a.ntiles = (int)nt;
    const int n_cus = device_cus();
    if (cin == 3) {
        hipLaunchKernelGGL(conv3x3_c3_kernel, dim3(a.ntiles < 4 * n_cus ? a.ntiles : 4 * n_cus), dim3(256), 0, stream, a);   // 21 KB of LDS, ~100 VGPRs
        DSRG_LAUNCH_CHECK();
        return DSRG_OK;
    }
    // 64 -> 64: one output tile per wave (144 weight VGPRs) and two workgroups per CU by default — the halo loads of a
    // workgroup are its only memory parallelism (24 KB in flight), a second one doubles it; DSRG_CONV_OCC=1: two tiles per wave
    static const bool occ2 = [] { const char *e = getenv("DSRG_CONV_OCC"); return !(e && !strcmp(e, "1")); }();
    int rc, grid;
    // (the masked form of the two-workgroup variant spills a few loop-invariant addresses — 281 against 267 us for conv1_2's data
    // gradient + the separate pass; two tiles per wave does not)
    if (cin == 64 && cout == 64 && occ2 && !mask) { rc = launch_variant<64, 64, 1>(a, n_cus, stream); grid = variant_grid<64, 64, 1>(a, n_cus); }
    else if (cin == 64 && cout == 64) { rc = launch_variant<64, 64, 2>(a, n_cus, stream); grid = variant_grid<64, 64, 2>(a, n_cus); }
    else if (cin == 64) { rc = launch_variant<64, 128, 2>(a, n_cus, stream); grid = variant_grid<64, 128, 2>(a, n_cus); }
    else if (cout == 64) { rc = launch_variant<128, 64, 1>(a, n_cus, stream); grid = variant_grid<128, 64, 1>(a, n_cus); }
    else { rc = launch_variant<128, 128, 1>(a, n_cus, stream); grid = variant_grid<128, 128, 1>(a, n_cus); }
    if (rc || !colsum) return rc;
    const float *parts[1] = {a.colsum};
    float *outs[1] = {colsum};
    return launch_igemm_colsum(parts, outs, 1, grid, cout, stream);
}

}  // namespace dsrg
