// Implicit-GEMM 3x3 (dilated) / 1x1 convolutions for the wide layers of the backbone: conv3_x at 81x81, conv4_x / conv5_x
// (dilation 1 / 2) and the four fc6_k (dilation 6 / 12 / 18 / 24) / fc7_k at 41x41 of train-s.prototxt:161-736 — backbone
// plumbing, no reference counterpart (Caffe's Convolution layer lives in the external framework).  One kernel serves the
// forward (bias + ReLU in the epilogue) and the data gradient (the same convolution of g with the flipped kernel, channel axes
// swapped: the caller packs the weights that way).
//
// Why it exists: the im2col route writes and re-reads a 9x copy of every activation (248 MB per 512-channel layer at batch 16,
// 1.2 ms per step in all) only so that a library GEMM can find its A operand contiguous.  Here the A operand is gathered:
//
//   out[m][n] = sum over (cc, tap, c) of  x[pixel(m) + tap offset][cc*64 + c] * w[n][cc][tap][c]
//
//   * tile 256 pixels x 256 output channels x 64 reduction elements (one tap of one 64-channel chunk per K-step), 8 waves
//     (2 across the channels x 4 across the pixels), v_mfma_f32_32x32x16_bf16, product taken transposed (rows = output
//     channels, columns = pixels) so that a lane ends with runs of four consecutive channels of ONE pixel;
//   * both operand tiles go global -> LDS by `buffer_load_dwordx4 ... lds` (no staging registers, no ds_write pass): a wave
//     instruction moves 8 rows of 128 bytes; the per-lane SOURCE offset is free, so the gather of the pixel rows (one
//     128-byte line per pixel, shifted by the tap's offset) costs one v_add + one v_cndmask per load, and a pixel whose tap
//     falls outside the map gets an offset beyond the descriptor's range, which the hardware answers with zeros;
//   * LDS rows are 128 bytes = 8 chunks of 16 bytes; chunk c of row r is stored at chunk position c ^ ((r >> 1) & 7)
//     (permuting the SOURCE chunk per lane — the LDS destination of the DMA is lane-linear), which spreads the 16 rows a
//     ds_read_b128 lane group touches over all 64 banks;
//   * K order: 64-channel chunk outer, tap inner — the nine taps re-read the same (shifted) 128-byte lines of a pixel
//     neighbourhood back to back, so they hit the XCD's L2; weights are packed [n][cc][tap][64] to match;
//   * two LDS stages of 64 KB: the DMA of K-step s + 1 is in flight while step s is multiplied, one barrier per step;
//   * up to four independent problems (the four ASPP branches: same geometry, own input / weights / dilation / output) share
//     a launch, so that the 424 tiles of one fc6 become 1696 and fill 256 CUs to 95 % instead of 83 %;
//   * epilogue: bias (+ ReLU) on the fp32 accumulators, bf16 pack, through LDS for 16-byte coalesced NHWC stores.
#include "common.h"
#include <cstdlib>
#include <cstring>

namespace dsrg {
namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int kBM = 256, kBN = 256, kBK = 64;
constexpr int kRow = kBK * 2;                              // bytes per LDS row
constexpr int kStage = (kBM + kBN) * kRow;                 // 64 KB: pixel rows, then weight rows
constexpr int kOutRow = 128 * 2 + 16;                      // epilogue: a wave's 64 pixels x 128 channels, padded rows
constexpr int kOutWave = 64 * kOutRow;
constexpr size_t kLdsBytes = (size_t)(2 * kStage > 8 * kOutWave ? 2 * kStage : 8 * kOutWave);
constexpr uint32_t kOob = 0x80000000u;                     // beyond any descriptor of this kernel: the load returns zeros

struct IgemmGroup {
    const uint16_t *x;      // (B, H, W, Cin) bf16
    const uint16_t *w;      // (Cout, Cin / 64, taps, 64) bf16
    const float *bias;      // (Cout) or nullptr
    uint16_t *y;            // (B, H, W, Cout) bf16
    int dil, pad_;
};
struct IgemmArgs {
    IgemmGroup g[4];
    int ngroups, B, H, W, Cin, Cout, taps, relu, M, tiles_m, tiles_n, tiles_per_group;
};

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {   // v_cvt_pk_bf16_f32: round to nearest even
    f32x2 v = {lo, hi};
    bf16x2 b = __builtin_convertvector(v, bf16x2);
    return *reinterpret_cast<uint32_t *>(&b);
}

// PREFETCH: the fragments of k-slice ks + 1 are read from LDS before the MFMAs of slice ks (order pinned by sched_barrier);
// otherwise the compiler places the reads (it sinks each next to its first use)
template <bool PREFETCH>
__global__ __launch_bounds__(512, 2) void conv_igemm_kernel(IgemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ig_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kgrp = lane >> 5;
    const int wn = wv >> 2, wm = wv & 3;

    // tile of this workgroup: consecutive ids share an XCD (blockIdx % 8) in runs, n-tile fastest
    int t;
    {
        const int total = (int)gridDim.x, id = (int)blockIdx.x;
        const int q = total >> 3, r = total & 7, xcd = id & 7, k = id >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int grp = t / a.tiles_per_group;
    t -= grp * a.tiles_per_group;
    const int tm = t / a.tiles_n, tn = t - tm * a.tiles_n;
    const int m0 = tm * kBM, n0 = tn * kBN;
    const IgemmGroup G = a.g[grp];
    const int taps = a.taps, Cin = a.Cin, W = a.W, H = a.H;
    const int ktot = taps * Cin;
    const int nsteps = (Cin >> 6) * taps;

    const rsrc_t rx = make_rsrc(G.x, (size_t)a.M * Cin * 2);
    const rsrc_t rw = make_rsrc(G.w, (size_t)a.Cout * ktot * 2);

    // ---- DMA geometry: per K-step a wave moves rows [wv*32 + i*8, +8) of both tiles, i = 0..3; lane -> (row, 16-byte chunk)
    uint32_t pbase[4], pvalid[4], wbase[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = wv * 32 + i * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);         // source chunk that lands at chunk position lane & 7
        const int m = m0 + r;
        const bool in = m < a.M;
        const int mm = in ? m : 0;
        const int hw = H * W;
        const int b = mm / hw, rem = mm - b * hw, y = rem / W, x = rem - y * W;
        uint32_t valid = 0;
        if (in) {
            if (taps == 9) {
#pragma unroll
                for (int tap = 0; tap < 9; tap++) {
                    const int yy = y + (tap / 3 - 1) * G.dil, xx = x + (tap % 3 - 1) * G.dil;
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) valid |= 1u << tap;
                }
            } else {
                valid = 1u;
            }
        }
        pbase[i] = (uint32_t)mm * (uint32_t)(Cin * 2) + (uint32_t)c * 16u;
        pvalid[i] = valid;
        wbase[i] = (uint32_t)(n0 + r) * (uint32_t)(ktot * 2) + (uint32_t)c * 16u;   // rows past Cout lie beyond the descriptor
    }

    auto issue = [&](int stage, int s) {
        const int cc = s / taps, tap = s - cc * taps;
        int dy = 0, dx = 0;
        if (taps == 9) { dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1; }
        const int toff = (dy * G.dil * W + dx * G.dil) * Cin * 2 + cc * kRow;          // wave-uniform
        unsigned char *P = ig_lds + stage * kStage + wv * (32 * kRow);
        unsigned char *Wt = P + kBM * kRow;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t vo = ((pvalid[i] >> tap) & 1u) ? pbase[i] + (uint32_t)toff : kOob;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void *)(P + i * (8 * kRow)), 16, vo, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)(Wt + i * (8 * kRow)), 16, wbase[i], (uint32_t)s * kRow, 0, 0);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    // per-lane LDS read offsets: row l31 of a 32-row fragment, chunk (ks*2 + kgrp) ^ swizzle(row); the swizzle term
    // ((row >> 1) & 7) depends on the lane only (fragment bases are multiples of 32 rows)
    const int sw = (l31 >> 1) & 7;
    const uint32_t rowoff = (uint32_t)l31 * kRow;
    uint32_t choff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) choff[ks] = rowoff + (uint32_t)(((ks * 2 + kgrp) ^ sw) << 4);

    issue(0, 0);
    for (int s = 0; s < nsteps; s++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of step s has landed
        __syncthreads();                                    // ... everybody's has, and everybody is done with step s - 1
        if (s + 1 < nsteps) issue((s + 1) & 1, s + 1);
        const unsigned char *P = ig_lds + (s & 1) * kStage + wm * (64 * kRow);
        const unsigned char *Wt = ig_lds + (s & 1) * kStage + kBM * kRow + wn * (128 * kRow);
        if (PREFETCH) {
            bf16x8 af[2][4], bfr[2][2];
#pragma unroll
            for (int i = 0; i < 4; i++) af[0][i] = *reinterpret_cast<const bf16x8 *>(Wt + i * (32 * kRow) + choff[0]);
#pragma unroll
            for (int j = 0; j < 2; j++) bfr[0][j] = *reinterpret_cast<const bf16x8 *>(P + j * (32 * kRow) + choff[0]);
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                if (ks + 1 < 4) {
#pragma unroll
                    for (int i = 0; i < 4; i++) af[(ks + 1) & 1][i] = *reinterpret_cast<const bf16x8 *>(Wt + i * (32 * kRow) + choff[(ks + 1) & 3]);
#pragma unroll
                    for (int j = 0; j < 2; j++) bfr[(ks + 1) & 1][j] = *reinterpret_cast<const bf16x8 *>(P + j * (32 * kRow) + choff[(ks + 1) & 3]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][i], bfr[ks & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                bf16x8 af[4], bfr[2];
#pragma unroll
                for (int i = 0; i < 4; i++) af[i] = *reinterpret_cast<const bf16x8 *>(Wt + i * (32 * kRow) + choff[ks]);
#pragma unroll
                for (int j = 0; j < 2; j++) bfr[j] = *reinterpret_cast<const bf16x8 *>(P + j * (32 * kRow) + choff[ks]);
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                        // every wave is done reading the last stage

    // ---- epilogue: C[row = channel][col = pixel]; a lane holds channels (reg & 3) + 8 (reg >> 2) + 4 kgrp of pixel l31
    unsigned char *O = ig_lds + wv * kOutWave;
    const int nw = n0 + wn * 128;
    const rsrc_t rb = make_rsrc(G.bias, G.bias ? (size_t)a.Cout * 4 : 0);      // no bias: every load is out of range = 0
    const float floor_ = a.relu ? 0.0f : -__builtin_inff();                    // ReLU without a branch per value
    float bias_r[4][4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int e = 0; e < 4; e++)
                bias_r[i][q][e] = ld_f32(rb, (uint32_t)(nw + i * 32 + q * 8 + kgrp * 4 + e) * 4u);
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int nl = i * 32 + q * 8 + kgrp * 4;
            const float4 b4 = make_float4(bias_r[i][q][0], bias_r[i][q][1], bias_r[i][q][2], bias_r[i][q][3]);
#pragma unroll
            for (int j = 0; j < 2; j++) {
                float v0 = acc[i][j][q * 4 + 0] + b4.x, v1 = acc[i][j][q * 4 + 1] + b4.y;
                float v2 = acc[i][j][q * 4 + 2] + b4.z, v3 = acc[i][j][q * 4 + 3] + b4.w;
                v0 = fmaxf(v0, floor_); v1 = fmaxf(v1, floor_); v2 = fmaxf(v2, floor_); v3 = fmaxf(v3, floor_);
                *reinterpret_cast<uint2 *>(O + (j * 32 + l31) * kOutRow + nl * 2) = make_uint2(pack2(v0, v1), pack2(v2, v3));
            }
        }
    }
    // the wave reads back its own rows only: LDS operations of one wave complete in order
#pragma unroll 4
    for (int it = 0; it < 16; it++) {
        const int p = it * 4 + (lane >> 4), ch = lane & 15;
        const int m = m0 + wm * 64 + p;
        const uint4 v = *reinterpret_cast<const uint4 *>(O + p * kOutRow + ch * 16);
        if (m < a.M) *reinterpret_cast<uint4 *>(G.y + (size_t)m * a.Cout + nw + ch * 8) = v;
    }
}

}  // namespace

bool conv_igemm_supported(int cin, int cout, int k) {
    return (k == 1 || k == 3) && cin >= 64 && cin % 64 == 0 && cout >= kBN && cout % kBN == 0;
}

int g_igemm_variant = -1;      // dsrg_debug_set_igemm_variant (tests / tools); -1 = DSRG_IGEMM_VARIANT or the default
static int igemm_variant() {
    if (g_igemm_variant < 0) { const char *e = getenv("DSRG_IGEMM_VARIANT"); g_igemm_variant = e ? atoi(e) : 1; }
    return g_igemm_variant;
}

int launch_conv_igemm(const void *const *x, const void *const *w, const float *const *bias, void *const *y, const int *dil,
                      int ngroups, int B, int H, int W, int cin, int cout, int k, int relu, hipStream_t stream) {
    if (ngroups < 1 || ngroups > 4) return set_error(DSRG_ERR_INVALID, "conv_igemm: 1..4 groups");
    if (!conv_igemm_supported(cin, cout, k))
        return set_error(DSRG_ERR_UNSUPPORTED, "conv_igemm: cin %% 64 == 0, cout %% %d == 0, k in (1, 3) required (got %d, %d, %d)",
                         kBN, cin, cout, k);
    const long long M = (long long)B * H * W;
    if (M <= 0 || M * cin * 2 >= 0x7fffffffLL || (long long)cout * k * k * cin * 2 >= 0x7fffffffLL || M * cout * 2 >= 0x7fffffff00LL)
        return set_error(DSRG_ERR_UNSUPPORTED, "conv_igemm: tensor too large for 32-bit buffer offsets");
    IgemmArgs a;
    memset(&a, 0, sizeof(a));
    for (int g = 0; g < ngroups; g++) {
        a.g[g].x = static_cast<const uint16_t *>(x[g]);
        a.g[g].w = static_cast<const uint16_t *>(w[g]);
        a.g[g].bias = bias ? bias[g] : nullptr;
        a.g[g].y = static_cast<uint16_t *>(y[g]);
        a.g[g].dil = dil ? dil[g] : 1;
        if (!a.g[g].x || !a.g[g].w || !a.g[g].y) return set_error(DSRG_ERR_INVALID, "conv_igemm: null pointer");
    }
    a.ngroups = ngroups; a.B = B; a.H = H; a.W = W; a.Cin = cin; a.Cout = cout; a.taps = k * k; a.relu = relu; a.M = (int)M;
    a.tiles_m = (int)((M + kBM - 1) / kBM);
    a.tiles_n = cout / kBN;
    a.tiles_per_group = a.tiles_m * a.tiles_n;
    static LdsGrant grant[2];
    const int variant = igemm_variant() ? 1 : 0;
    const void *fn = variant ? reinterpret_cast<const void *>(&conv_igemm_kernel<true>) : reinterpret_cast<const void *>(&conv_igemm_kernel<false>);
    if (int rc = ensure_dynamic_lds(fn, kLdsBytes, grant[variant])) return rc;
    if (variant) hipLaunchKernelGGL(conv_igemm_kernel<true>, dim3(a.tiles_per_group * ngroups), dim3(512), kLdsBytes, stream, a);
    else hipLaunchKernelGGL(conv_igemm_kernel<false>, dim3(a.tiles_per_group * ngroups), dim3(512), kLdsBytes, stream, a);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

}  // namespace dsrg
