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
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <queue>
#include <vector>

namespace dsrg {
namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int kBM = 256, kBN = 256;
constexpr int kOutRow = 128 * 2 + 16;                      // epilogue: a wave's 64 pixels x 128 channels, padded rows
constexpr int kOutWave = 64 * kOutRow;
constexpr uint32_t kOob = 0x80000000u;                     // beyond any descriptor of this kernel: the load returns zeros
constexpr int kRowTab = 8 * kOutWave;                      // cls_tiles: the pixel of each of the tile's 256 rows (int32, -1 = none), behind
constexpr int kRowTabBytes = kBM * 4;                      // the stages and the epilogue's staging rows

// BK = reduction elements per K-step, NST = LDS stages.  (64, 2): one step in flight behind the one being multiplied;
// (32, 4): a ring with three steps in flight (counted vmcnt: the DMA of steps s + 1, s + 2 stays in flight across the barrier of
// step s) — same 128 KB of LDS, deeper prefetch, twice the barriers.
__device__ __forceinline__ void wait_vm_lgkm_barrier() {
    // as wait_vm_barrier<0>, and this wave's LDS reads have returned too (the barrier then also says "done reading")
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int BK, int NST> struct ICfg {
    static constexpr int ROW = BK * 2;                     // bytes per LDS row
    static constexpr int CPR = ROW / 16;                   // 16-byte chunks per row (8 / 4)
    static constexpr int RPI = 1024 / ROW;                 // rows one wave DMA instruction moves (8 / 16)
    static constexpr int IPW = 256 / RPI / 8;              // DMA instructions per wave, tile and step (4 / 2)
    static constexpr int STAGE = (kBM + kBN) * ROW;        // pixel rows, then weight rows (64 / 32 KB)
    static constexpr int KS = BK / 16;                     // MFMA k-slices per step
    static constexpr int SPC = 64 / BK;                    // steps per 64-channel chunk of a tap
    static constexpr int AHEAD = NST - 1;
    static_assert(NST * STAGE <= kRowTab, "the row table sits behind the stages");
    static constexpr size_t LDS = (size_t)kRowTab + kRowTabBytes;
    // chunk c of row r is stored at chunk position c ^ swz(r): the 16 rows a ds_read_b128 lane group touches then cover
    // all 64 banks (rows of 128 B: 2 rows per 256-byte bank line -> 3 bits from r >> 1; rows of 64 B: 4 per line -> r >> 2)
    __device__ static constexpr int swz(int r) { return BK == 64 ? (r >> 1) & 7 : (r >> 2) & 3; }
};

// a rectangle [y0, y0 + h) x [x0, x0 + w) of every image of the batch; its pixels (image-major, raster order inside) take the
// indices q0 .. q0 + B h w - 1 of the launch's pixel order (IgemmArgs::cls_tiles)
struct IgemmClass {
    uint32_t rect;          // y0 | h << 8 | x0 << 16 | w << 24
    int q0;
};
constexpr int kMaxClasses = 9;
struct IgemmGroup {
    const uint16_t *x;      // (B, H, W, Cin) bf16
    const uint16_t *w;      // (Cout, Cin / 64, taps, 64) bf16
    const float *bias;      // (Cout) or nullptr
    uint16_t *y;            // (B, H, W, Cout) bf16
    const uint16_t *mask;   // (B, H, W, Cout) bf16 or nullptr: outputs are kept where mask > 0 (times out_scale), else zeroed —
                            // the ReLU (+ Dropout) backward of the layer below, when this launch is a data gradient
    float *colsum;          // (tiles_m, Cout) fp32 or nullptr: per 256-row tile, the column sums of what was stored —
                            // the partial bias gradient of the layer below (summed in fixed order by igemm_colsum_kernel)
    const uint16_t *res;    // (B, H, W, Cout) bf16 or nullptr: added to the bf16-rounded result BEFORE the ReLU / the mask — the identity
                            // branch of a residual block in the forward launch (y = relu(bf16(conv + bias) + res): bit for bit what a separate
                            // add + ReLU pass over the stored convolution gives), the gradient that reaches the block's input past the
                            // convolutions in its first convolution's data gradient (gx = (bf16(conv) + res) where mask > 0)
    int dil, ncls;
    IgemmClass cls[kMaxClasses];    // cls_tiles: the group's pixel order, classes by ascending q0
};
struct IgemmArgs {
    IgemmGroup g[4];
    int ngroups, B, H, W, Cin, Cout, taps, relu, M, tiles_m, tiles_n, tiles_per_group;
    int stagger;            // 1: waves 4-7 issue their DMA behind the first MFMA cluster of a step (their SIMD partners 0-3 issue
                            // in front of theirs), so that no SIMD's matrix pipe waits for both of its waves to get through the issue code
    uint32_t drop_thresh;   // Dropout behind the ReLU, fused: keep an element iff its random byte >= drop_thresh (p = thresh / 256;
    float drop_scale;       // 0 = no dropout), kept elements times drop_scale = 1 / (1 - p)
    uint32_t seed_lo, seed_hi;
    float out_scale;        // every output times this (1 unless a mask carries a Dropout scale)
    int xcd_mix;            // 1: XCD-interleaved tile map (see the kernel); the grid is 8 * ngroups * ceil(tiles_m / 8) * tiles_n blocks
    int row_tiles, rows_per_tile, bands;      // 1: a pixel tile = rows_per_tile whole rows of one image (bands of them per image)
    int skip_taps;          // 1: a K-step whose tap reaches no pixel of the tile (a dilated kernel near the map's border: all of
                            // its operand rows would be the zeros of the padding) is not loaded and not multiplied
    int xrow;               // bytes per pixel row of x (Cin * 2 — or, split mode, the three bf16 planes of a float32 activation: 3 * cin * 2)
    int cpp;                // split mode (0 = off): 64-channel chunks per PLANE.  A float32 convolution as six bf16 products on the fp32
                            // accumulators: x = x0 + x1 + x2 and w = w0 + w1 + w2 (each term the bf16 rounding of what the terms before
                            // leave), the products x0 w0, x0 w1, x1 w0, x0 w2, x2 w0, x1 w1 carry 2^-24-grade relative error; the K loop
                            // runs over 6 * cpp VIRTUAL chunks — virtual chunk v reads plane kSplitXPlane[v / cpp] of x, and the packed
                            // kernel holds plane kSplitWPlane[v / cpp] of w there (the caller's packing)
    int out_f32;            // 1: y is float32 (B, H, W, Cout), written straight from the accumulators (split mode's prototype epilogue)
    int cls_tiles;          // 1: a pixel tile = 256 consecutive indices of the group's CLASS order (IgemmGroup::cls) instead of 256
                            // consecutive pixels: a dilated tap (dy, dx) d reaches exactly the pixels of a rectangle of the map, so the
                            // map falls into <= 3 x 3 rectangles inside each of which every pixel has the SAME live taps; ordered
                            // class by class (most taps first), a tile multiplies padding only where it straddles two classes
};

// the random bytes of the four consecutive channels starting at element 4 * e4 of a launch's output: a counter-based
// generator (two rounds of the murmur3 finaliser over the element counter and the 64-bit seed) — every element's decision is a
// pure function of (seed, position), so the mask needs no state, no extra pass and no storage (the backward pass reads it
// off the sign of the output, as with torch's dropout kernel before)
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ uint32_t dropout_bytes(uint32_t e4, uint32_t seed_lo, uint32_t seed_hi) {
    return fmix32(fmix32(e4 ^ seed_lo) + seed_hi);
}

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {   // v_cvt_pk_bf16_f32: round to nearest even
    f32x2 v = {lo, hi};
    bf16x2 b = __builtin_convertvector(v, bf16x2);
    return *reinterpret_cast<uint32_t *>(&b);
}

template <int N> __device__ __forceinline__ void wait_vm_barrier() {
    // this wave's DMA up to the step about to be read has landed (N younger loads may still fly), then the workgroup barrier:
    // everybody's has, and everybody is done reading the stage the next issue overwrites.  One asm statement with a memory
    // clobber: neither LDS reads nor the DMA issue may move across it.
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// (the body is a device function template behind two plain kernels: hipcc 7.2's HOST pass drops the stub of a kernel TEMPLATE
// whose body calls a lambda that uses the template's constants — no diagnostic, an undefined symbol at load time)
// EARLY (two stages only): the step's barrier sits in front of its LAST MFMA cluster instead of behind it, and the first
// fragments of the next step are read (from the other stage, complete and visible once the barrier is passed) before that
// cluster — so the LDS round trip of a step's first fragments runs under MFMAs instead of holding both waves of a SIMD right
// behind the barrier.  All fragment reads of the current stage have been issued by then (the last slice is prefetched during
// the one before) and are waited for in front of the barrier, so the stage is free for the DMA of step s + 2 as before.
template <int BK, int NST, bool EARLY = false>
__device__ __forceinline__ void conv_igemm_body(const IgemmArgs &a, const int bid, const int nblk) {      // block bid of nblk (a merged launch passes a sub-range)
    using C = ICfg<BK, NST>;
    extern __shared__ __attribute__((aligned(16))) unsigned char ig_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kgrp = lane >> 5;
    const int wn = wv >> 2, wm = wv & 3;

    // tile of this workgroup: consecutive ids share an XCD (blockIdx % 8) in runs, n-tile fastest.
    // xcd_mix (launches that skip taps): tiles no longer cost the same — a dilation-24 tile runs 3 to 6 of its 9 taps, a
    // dilation-6 tile all of them — and an XCD only ever runs the blocks with its own id % 8: with runs of consecutive tiles
    // per XCD the two XCDs that hold fc6_1's tiles set the launch's duration while those with fc6_4's idle (measured: skipping
    // 15 % of the steps bought 0 %).  There XCD x takes the pixel tiles x, x + 8, .. of EVERY group (its blocks in the
    // order group, pixel tile, n-tile), so that all XCDs hold the same mix; blocks beyond an XCD's share exit at once.
    int grp, tm, tn;
    if (a.xcd_mix) {
        const int id = bid, xcd = id & 7, k = id >> 3;
        const int cm = (a.tiles_m - xcd + 7) >> 3, per_group = cm * a.tiles_n;
        if (cm <= 0 || k >= per_group * a.ngroups) return;
        grp = k / per_group;
        const int r = k - grp * per_group, tml = r / a.tiles_n;
        tn = r - tml * a.tiles_n;
        tm = xcd + 8 * tml;
    } else {
        const int total = nblk, id = bid;
        const int q = total >> 3, r = total & 7, xcd = id & 7, k = id >> 3;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
        grp = t / a.tiles_per_group;
        t -= grp * a.tiles_per_group;
        tm = t / a.tiles_n;
        tn = t - tm * a.tiles_n;
    }
    // the tile's pixels: 256 consecutive ones — or (row_tiles: launches that skip taps) whole map rows of ONE image, as many as fit
    // 256: a tile that straddles two images reaches every vertical tap through one of them and skips nothing, a tile aligned to
    // rows skips more (41-wide map, six rows = 246 pixels per tile: 5 % fewer K-steps over the four fc6_k than flattened tiles)
    int m0 = tm * kBM, mvalid;
    if (a.row_tiles) {
        const int bimg = tm / a.bands, band = tm - bimg * a.bands, row0 = band * a.rows_per_tile;
        m0 = (bimg * a.H + row0) * a.W;
        mvalid = min(a.rows_per_tile, a.H - row0) * a.W;
    } else {
        mvalid = min(kBM, a.M - m0);
    }
    const int n0 = tn * kBN;
    const IgemmGroup &G = a.g[grp];
    const int taps = a.taps, Cin = a.Cin, W = a.W, H = a.H;
    const int ktot = taps * Cin;
    // cls_tiles: row r of the tile is index m0 + r of the group's class order; its pixel goes to the row table (one row per
    // thread: a search over <= 9 classes and two divisions), which the DMA geometry and the epilogue read
    volatile int32_t *rowpix = reinterpret_cast<volatile int32_t *>(ig_lds + kRowTab);
    const bool cls = a.cls_tiles != 0;
    if (cls) {
        if (tid < kBM) {
            const int q = m0 + tid;
            int m = -1;
            if (q < a.M) {
                uint32_t rect = G.cls[0].rect;
                int q0 = G.cls[0].q0;
#pragma unroll
                for (int k = 1; k < kMaxClasses; k++)
                    if (k < G.ncls && q >= G.cls[k].q0) { rect = G.cls[k].rect; q0 = G.cls[k].q0; }
                const int cy0 = (int)(rect & 255u), ch = (int)((rect >> 8) & 255u), cx0 = (int)((rect >> 16) & 255u), cw = (int)(rect >> 24);
                const int qq = q - q0, area = ch * cw, b = qq / area, rem = qq - b * area, yy = rem / cw, xx = rem - yy * cw;
                m = (b * H + cy0 + yy) * W + cx0 + xx;
            }
            rowpix[tid] = m;
        }
        __syncthreads();
    }
    auto row_pixel = [&](int r) -> int { return cls ? rowpix[r] : (r < mvalid ? m0 + r : -1); };      // -1: the tile has no such row

    const int xrow = a.xrow;
    const rsrc_t rx = make_rsrc(G.x, (size_t)a.M * xrow);
    const rsrc_t rw = make_rsrc(G.w, (size_t)a.Cout * ktot * 2);

    // ---- DMA geometry: per K-step a wave moves rows [wv*32 + i*RPI, +RPI) of both tiles; lane -> (row, 16-byte chunk)
    uint32_t pbase[C::IPW], pvalid[C::IPW], wbase[C::IPW];
#pragma unroll
    for (int i = 0; i < C::IPW; i++) {
        const int r = wv * 32 + i * C::RPI + lane / C::CPR;
        const int c = (lane % C::CPR) ^ C::swz(r);         // source chunk that lands at chunk position lane % CPR
        const int m = row_pixel(r);
        const bool in = m >= 0;
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
        pbase[i] = (uint32_t)mm * (uint32_t)xrow + (uint32_t)c * 16u;
        pvalid[i] = valid;
        wbase[i] = (uint32_t)(n0 + r) * (uint32_t)(ktot * 2) + (uint32_t)c * 16u;   // rows past Cout lie beyond the descriptor
    }

    // ---- the taps this tile multiplies.  With a dilated kernel whole taps fall into the zero padding for every pixel of a tile
    // (fc6_4, dilation 24 on a 41-row map: the three upper taps reach a pixel only from row 24 on — for the tiles above they
    // are 64-deep K-steps of zeros times weights): the tile's taps are the OR of its pixels' validity bits, and the K loop runs
    // over those only — 15 % of the four fc6_k's steps at 16 images (30 % for dilation 24).  A skipped step adds exact zeros
    // to every accumulator, so results are bit-identical with and without (tests/test_gpu_igemm.py).  Taps stay in their
    // order, chunk outer / tap inner as before; a tile that reaches every tap runs the very same sequence.
    uint32_t tapmask = taps == 9 ? 0x1ffu : 1u;
    if (a.skip_taps && taps == 9 && G.dil >= 3) {           // (workgroup-uniform; below 3 no 256-pixel tile loses a tap)
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < C::IPW; i++) v |= pvalid[i];
#pragma unroll
        for (int o = 32; o; o >>= 1) v |= (uint32_t)__shfl_xor((int)v, o);
        volatile uint32_t *red = reinterpret_cast<volatile uint32_t *>(ig_lds);
        if (lane == 0) red[wv] = v;
        __syncthreads();
        v = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) v |= red[k];
        __syncthreads();                                    // the words are stage 0's first row again
        tapmask = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    }
    unsigned long long taplist = 0;                         // nibble k = the k-th live tap
    int nlive = 0;
#pragma unroll
    for (int tp = 0; tp < 9; tp++)
        if ((tapmask >> tp) & 1u) { taplist |= (unsigned long long)tp << (4 * nlive); nlive++; }
    const int nsteps = (Cin >> 6) * nlive * C::SPC;

    // issue(stage, s): the DMA of K-step s into `stage`; the steps are issued in order, so (chunk, live tap, half) advance as
    // counters instead of being divided out of s
    int it_cc = 0, it_tap = 0, it_h = 0;
    auto issue = [&](int stage, int) {
        const int tap = (int)((taplist >> (4 * it_tap)) & 15u);
        const int cc = it_cc, h = it_h;
        const uint32_t s_real = (uint32_t)((cc * taps + tap) * C::SPC + h);     // position of the step's weights in a packed row
        if (++it_h == C::SPC) { it_h = 0; if (++it_tap == nlive) { it_tap = 0; it_cc++; } }
        int dy = 0, dx = 0;
        if (taps == 9) { dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1; }
        // (split mode: virtual chunk cc -> chunk cc % cpp of plane 0 0 1 0 2 1 [cc / cpp] of x; two bits per entry of 0x610)
        const int xc = a.cpp ? (int)((0x610u >> (2 * (cc / a.cpp))) & 3u) * a.cpp + cc % a.cpp : cc;
        const int toff = (dy * G.dil * W + dx * G.dil) * xrow + xc * 128 + h * C::ROW;      // wave-uniform
        unsigned char *P = ig_lds + stage * C::STAGE + wv * (32 * C::ROW);
        unsigned char *Wt = P + kBM * C::ROW;
#pragma unroll
        for (int i = 0; i < C::IPW; i++) {
            const uint32_t vo = ((pvalid[i] >> tap) & 1u) ? pbase[i] + (uint32_t)toff : kOob;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void *)(P + i * (C::RPI * C::ROW)), 16, vo, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < C::IPW; i++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)(Wt + i * (C::RPI * C::ROW)), 16, wbase[i], s_real * C::ROW, 0, 0);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    // per-lane LDS read offsets: row l31 of a 32-row fragment, chunk (ks*2 + kgrp) ^ swizzle(row); the swizzle term depends on
    // the lane only (fragment bases are multiples of 32 rows)
    const int sw = C::swz(l31);
    const uint32_t rowoff = (uint32_t)l31 * C::ROW;
    uint32_t choff[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ks++) choff[ks] = rowoff + (uint32_t)(((ks * 2 + kgrp) ^ sw) << 4);

    // an output-channel count of 128 (mod 256) leaves the upper half of the last n-tile empty: its waves (4-7) only move data
    const bool active = n0 + wn * 128 < a.Cout;
    constexpr int LPS = 2 * C::IPW;                         // DMA instructions per wave and step
    if (EARLY) {
        static_assert(!EARLY || NST == 2, "the early-barrier schedule is written for two stages");
        bf16x8 af[2][4], bfr[2][2];
        const bool late = a.stagger && wv >= 4 && active;
        issue(0, 0);
        if (nsteps > 1) { issue(1, 1); wait_vm_barrier<LPS>(); } else { wait_vm_barrier<0>(); }
        if (active) {
            const unsigned char *P = ig_lds + wm * (64 * C::ROW), *Wt = ig_lds + kBM * C::ROW + wn * (128 * C::ROW);
#pragma unroll
            for (int i = 0; i < 4; i++) af[0][i] = *reinterpret_cast<const bf16x8 *>(Wt + i * (32 * C::ROW) + choff[0]);
#pragma unroll
            for (int j = 0; j < 2; j++) bfr[0][j] = *reinterpret_cast<const bf16x8 *>(P + j * (32 * C::ROW) + choff[0]);
        }
        for (int s = 0; s < nsteps; s++) {
            const int stage = s & 1;
            const bool next = s + 1 < nsteps, more = s + 2 < nsteps;
            if (!active) {                                  // a wave without output columns: it only moves data
                if (next) { wait_vm_lgkm_barrier(); if (more) issue(stage, s + 2); }
                continue;
            }
            const unsigned char *P = ig_lds + stage * C::STAGE + wm * (64 * C::ROW);
            const unsigned char *Wt = ig_lds + stage * C::STAGE + kBM * C::ROW + wn * (128 * C::ROW);
            const unsigned char *Pn = ig_lds + (stage ^ 1) * C::STAGE + wm * (64 * C::ROW);
            const unsigned char *Wn = ig_lds + (stage ^ 1) * C::STAGE + kBM * C::ROW + wn * (128 * C::ROW);
#pragma unroll
            for (int ks = 0; ks < C::KS; ks++) {
                if (ks + 1 < C::KS) {
#pragma unroll
                    for (int i = 0; i < 4; i++) af[(ks + 1) & 1][i] = *reinterpret_cast<const bf16x8 *>(Wt + i * (32 * C::ROW) + choff[ks + 1 < C::KS ? ks + 1 : 0]);
#pragma unroll
                    for (int j = 0; j < 2; j++) bfr[(ks + 1) & 1][j] = *reinterpret_cast<const bf16x8 *>(P + j * (32 * C::ROW) + choff[ks + 1 < C::KS ? ks + 1 : 0]);
                } else if (next) {
                    wait_vm_lgkm_barrier();                 // step s + 1 has landed for everybody; nobody reads this stage any more
                    if (more && !late) issue(stage, s + 2);
#pragma unroll
                    for (int i = 0; i < 4; i++) af[C::KS & 1][i] = *reinterpret_cast<const bf16x8 *>(Wn + i * (32 * C::ROW) + choff[0]);
#pragma unroll
                    for (int j = 0; j < 2; j++) bfr[C::KS & 1][j] = *reinterpret_cast<const bf16x8 *>(Pn + j * (32 * C::ROW) + choff[0]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][i], bfr[ks & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (ks == C::KS - 1 && more && late) issue(stage, s + 2);
            }
        }
    } else {
#pragma unroll
    for (int p = 0; p < C::AHEAD; p++)
        if (p < nsteps) issue(p, p);
    int stage = 0;
    for (int s = 0; s < nsteps; s++) {        // the steps behind s that are already on their way: min(AHEAD - 1, nsteps - 1 - s)
        if (C::AHEAD >= 3 && s + 2 < nsteps) wait_vm_barrier<2 * LPS>();
        else if (C::AHEAD >= 2 && s + 1 < nsteps) wait_vm_barrier<LPS>();
        else wait_vm_barrier<0>();
        const bool more = s + C::AHEAD < nsteps, late = a.stagger && wv >= 4 && active;
        const int nstage = stage == 0 ? NST - 1 : stage - 1;                                  // the stage step s - 1 was read from
        if (more && !late) issue(nstage, s + C::AHEAD);
        if (active) {
        const unsigned char *P = ig_lds + stage * C::STAGE + wm * (64 * C::ROW);
        const unsigned char *Wt = ig_lds + stage * C::STAGE + kBM * C::ROW + wn * (128 * C::ROW);
        // fragments of k-slice ks + 1 are read before the MFMAs of slice ks (order pinned by sched_barrier)
        bf16x8 af[2][4], bfr[2][2];
#pragma unroll
        for (int i = 0; i < 4; i++) af[0][i] = *reinterpret_cast<const bf16x8 *>(Wt + i * (32 * C::ROW) + choff[0]);
#pragma unroll
        for (int j = 0; j < 2; j++) bfr[0][j] = *reinterpret_cast<const bf16x8 *>(P + j * (32 * C::ROW) + choff[0]);
#pragma unroll
        for (int ks = 0; ks < C::KS; ks++) {
            if (ks + 1 < C::KS) {
#pragma unroll
                for (int i = 0; i < 4; i++) af[(ks + 1) & 1][i] = *reinterpret_cast<const bf16x8 *>(Wt + i * (32 * C::ROW) + choff[(ks + 1) % C::KS]);
#pragma unroll
                for (int j = 0; j < 2; j++) bfr[(ks + 1) & 1][j] = *reinterpret_cast<const bf16x8 *>(P + j * (32 * C::ROW) + choff[(ks + 1) % C::KS]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][i], bfr[ks & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 0 && more && late) issue(nstage, s + C::AHEAD);
        }
        }
        stage = stage + 1 == NST ? 0 : stage + 1;
    }
    }
    __syncthreads();                                        // every wave is done reading the last stage
    if (!active) return;

    // ---- epilogue: C[row = channel][col = pixel]; a lane holds channels (reg & 3) + 8 (reg >> 2) + 4 kgrp of pixel l31
    unsigned char *O = ig_lds + wv * kOutWave;
    const int nw = n0 + wn * 128;
    if (a.out_f32) {
        // float32 result straight from the accumulators: 16 bytes per lane and (i, q, j) — 32-byte runs per pixel row (the layer-level
        // prototype of the split mode; a production epilogue would stage through LDS as the bf16 one does)
        float *y32 = reinterpret_cast<float *>(G.y);
        const float floor32 = a.relu ? 0.0f : -__builtin_inff();
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int m = row_pixel(wm * 64 + j * 32 + l31);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int n = nw + i * 32 + q * 8 + kgrp * 4;
                    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (G.bias) b4 = *reinterpret_cast<const float4 *>(G.bias + n);
                    const float4 v = make_float4(fmaxf(acc[i][j][q * 4 + 0] + b4.x, floor32), fmaxf(acc[i][j][q * 4 + 1] + b4.y, floor32),
                                                 fmaxf(acc[i][j][q * 4 + 2] + b4.z, floor32), fmaxf(acc[i][j][q * 4 + 3] + b4.w, floor32));
                    if (m >= 0) *reinterpret_cast<float4 *>(y32 + (size_t)m * a.Cout + n) = v;
                }
        }
        return;
    }
    const rsrc_t rb = make_rsrc(G.bias, G.bias ? (size_t)a.Cout * 4 : 0);      // no bias: every load is out of range = 0
    const float floor_ = (a.relu && !G.res) ? 0.0f : -__builtin_inff();        // ReLU without a branch per value (behind the residual, if any)
    const uint32_t seed_g = a.seed_lo + (uint32_t)grp * 0x9E3779B9u;              // every branch its own stream
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
                v0 = fmaxf(v0, floor_) * a.out_scale; v1 = fmaxf(v1, floor_) * a.out_scale;
                v2 = fmaxf(v2, floor_) * a.out_scale; v3 = fmaxf(v3, floor_) * a.out_scale;
                if (a.drop_thresh) {                                             // uniform
                    const uint32_t m = cls ? (uint32_t)rowpix[wm * 64 + j * 32 + l31] : (uint32_t)(m0 + wm * 64 + j * 32 + l31);
                    const uint32_t h = dropout_bytes((m * (uint32_t)a.Cout + (uint32_t)(nw + nl)) >> 2, seed_g, a.seed_hi);
                    v0 = (h & 0xffu) >= a.drop_thresh ? v0 * a.drop_scale : 0.0f;
                    v1 = ((h >> 8) & 0xffu) >= a.drop_thresh ? v1 * a.drop_scale : 0.0f;
                    v2 = ((h >> 16) & 0xffu) >= a.drop_thresh ? v2 * a.drop_scale : 0.0f;
                    v3 = (h >> 24) >= a.drop_thresh ? v3 * a.drop_scale : 0.0f;
                }
                *reinterpret_cast<uint2 *>(O + (j * 32 + l31) * kOutRow + nl * 2) = make_uint2(pack2(v0, v1), pack2(v2, v3));
            }
        }
    }
    // the wave reads back its own rows only: LDS operations of one wave complete in order
    const bool chv = nw + (lane & 15) * 8 < a.Cout;          // a 64-channel output: the upper half of the wave's 128 columns does not exist
    if (G.res) {                                             // (uniform) residual, then ReLU and / or the mask of the layer below
        const float floor2 = a.relu ? 0.0f : -__builtin_inff();
#pragma unroll
        for (int h = 0; h < 2; h++) {                        // two batches of eight rows: two latencies, half the registers
            uint4 rs[8], mk8[8];
#pragma unroll
            for (int it = 0; it < 8; it++) {
                const int m = row_pixel(wm * 64 + (h * 8 + it) * 4 + (lane >> 4));
                const size_t at = (size_t)m * a.Cout + nw + (lane & 15) * 8;
                rs[it] = (m >= 0 && chv) ? *reinterpret_cast<const uint4 *>(G.res + at) : make_uint4(0, 0, 0, 0);
                mk8[it] = !G.mask ? make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u)
                                  : ((m >= 0 && chv) ? *reinterpret_cast<const uint4 *>(G.mask + at) : make_uint4(0, 0, 0, 0));
            }
#pragma unroll
            for (int it = 0; it < 8; it++) {
                const int p = (h * 8 + it) * 4 + (lane >> 4), ch = lane & 15;
                const int m = row_pixel(wm * 64 + p);
                const uint4 v = *reinterpret_cast<const uint4 *>(O + p * kOutRow + ch * 16);
                uint32_t w4[4] = {v.x, v.y, v.z, v.w};
                const uint32_t r4[4] = {rs[it].x, rs[it].y, rs[it].z, rs[it].w};
                const uint32_t y4[4] = {mk8[it].x, mk8[it].y, mk8[it].z, mk8[it].w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float lo = fmaxf(__uint_as_float(w4[e] << 16) + __uint_as_float(r4[e] << 16), floor2);
                    const float hi = fmaxf(__uint_as_float(w4[e] & 0xffff0000u) + __uint_as_float(r4[e] & 0xffff0000u), floor2);
                    const uint32_t klo = (int16_t)(y4[e] & 0xffffu) > 0 ? 0x0000ffffu : 0u;
                    const uint32_t khi = (int32_t)y4[e] >= 0x10000 ? 0xffff0000u : 0u;
                    w4[e] = pack2(lo, hi) & (klo | khi);
                }
                if (m >= 0 && chv) *reinterpret_cast<uint4 *>(G.y + (size_t)m * a.Cout + nw + ch * 8) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            }
        }
        return;
    }
    if (!G.mask && !G.colsum) {
#pragma unroll 4
        for (int it = 0; it < 16; it++) {
            const int p = it * 4 + (lane >> 4), ch = lane & 15;
            const int m = row_pixel(wm * 64 + p);
            const uint4 v = *reinterpret_cast<const uint4 *>(O + p * kOutRow + ch * 16);
            if (m >= 0 && chv) *reinterpret_cast<uint4 *>(G.y + (size_t)m * a.Cout + nw + ch * 8) = v;
        }
        return;
    }
    // data gradient with the backward of the ReLU (+ Dropout) below it and that layer's bias gradient: keep a value where the
    // layer's OUTPUT was positive (bf16 halves compared as signed 16-bit integers: +0 and negatives drop), sum what is stored.
    // The accumulators are dead by now: all sixteen mask rows of the lane are fetched in one batch (one latency, not sixteen).
    uint4 mk[16];
    if (G.mask) {
#pragma unroll
        for (int it = 0; it < 16; it++) {
            const int m = row_pixel(wm * 64 + it * 4 + (lane >> 4));
            mk[it] = (m >= 0 && chv) ? *reinterpret_cast<const uint4 *>(G.mask + (size_t)m * a.Cout + nw + (lane & 15) * 8) : make_uint4(0, 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int it = 0; it < 16; it++) mk[it] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    }
    float cs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int it = 0; it < 16; it++) {
        const int p = it * 4 + (lane >> 4), ch = lane & 15;
        const int m = row_pixel(wm * 64 + p);
        const uint4 v = *reinterpret_cast<const uint4 *>(O + p * kOutRow + ch * 16);
        uint32_t w4[4] = {v.x, v.y, v.z, v.w};
        const uint32_t y4[4] = {mk[it].x, mk[it].y, mk[it].z, mk[it].w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t lo = (int16_t)(y4[e] & 0xffffu) > 0 ? 0x0000ffffu : 0u;
            const uint32_t hi = (int32_t)y4[e] >= 0x10000 ? 0xffff0000u : 0u;       // upper half positive and non-zero
            w4[e] &= lo | hi;                                                        // (rows past the end: mask 0 -> adds 0)
            cs[2 * e] += __uint_as_float(w4[e] << 16);
            cs[2 * e + 1] += __uint_as_float(w4[e] & 0xffff0000u);
        }
        if (m >= 0 && chv) *reinterpret_cast<uint4 *>(G.y + (size_t)m * a.Cout + nw + ch * 8) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    }
    if (G.colsum) {                                          // uniform for the workgroup
#pragma unroll
        for (int e = 0; e < 8; e++) {
            cs[e] += __shfl_xor(cs[e], 16);
            cs[e] += __shfl_xor(cs[e], 32);
        }
        // the four 64-row slabs of the tile are summed here (fixed order) so that the scratch holds one row per tile.  A wave's
        // own staging region is free again (its reads above have returned: their values were used)
        float *slab = reinterpret_cast<float *>(O);
        if (lane < 16) {
            *reinterpret_cast<float4 *>(slab + lane * 8) = make_float4(cs[0], cs[1], cs[2], cs[3]);
            *reinterpret_cast<float4 *>(slab + lane * 8 + 4) = make_float4(cs[4], cs[5], cs[6], cs[7]);
        }
        __syncthreads();                                     // (waves of a 128-wide n-tile's idle half have left: not counted)
        if (wm == 0) {
            const float *s0 = reinterpret_cast<const float *>(ig_lds + (wn * 4 + 0) * kOutWave);
            const float *s1 = reinterpret_cast<const float *>(ig_lds + (wn * 4 + 1) * kOutWave);
            const float *s2 = reinterpret_cast<const float *>(ig_lds + (wn * 4 + 2) * kOutWave);
            const float *s3 = reinterpret_cast<const float *>(ig_lds + (wn * 4 + 3) * kOutWave);
            float *dst = G.colsum + (size_t)tm * a.Cout + nw;
            dst[lane] = (s0[lane] + s1[lane]) + (s2[lane] + s3[lane]);
            if (nw + lane + 64 < a.Cout) dst[lane + 64] = (s0[lane + 64] + s1[lane + 64]) + (s2[lane + 64] + s3[lane + 64]);
        }
    }
}


__global__ __launch_bounds__(512, 2) void conv_igemm_kernel_64x2(IgemmArgs a) { conv_igemm_body<64, 2>(a, (int)blockIdx.x, (int)gridDim.x); }
__global__ __launch_bounds__(512, 2) void conv_igemm_kernel_64x2e(IgemmArgs a) { conv_igemm_body<64, 2, true>(a, (int)blockIdx.x, (int)gridDim.x); }
__global__ __launch_bounds__(512, 2) void conv_igemm_kernel_32x4(IgemmArgs a) { conv_igemm_body<32, 4>(a, (int)blockIdx.x, (int)gridDim.x); }

// ------------------------------------------------------------------------------------------------------------------
// The same convolution with the K-steps of ALL tiles dealt out evenly ("stream-K"): the 41x41 layers have 212 tiles of 72
// K-steps for 256 CUs — one round with a sixth of the chip idle — and the data gradient of conv4_1 106.  Here the grid is one
// workgroup per CU and workgroup u takes the global K-steps [u T / U, (u + 1) T / U) of the T = tiles x steps of the launch:
// up to one tile it finishes for an earlier workgroup (its first segment), whole tiles, and up to one tile it starts (its last
// segment).  A tile cut this way is completed by the workgroup that holds its FIRST step — that segment is the last thing the
// workgroup does, so the other parts (the first segments of the following workgroups, written to a workspace with
// write-through stores and published by a flag: cdna_hip_programming.md Guideline 16, recipe R1) are there long before it
// looks; it adds them in workgroup order (deterministic) and runs the ordinary epilogue.  Dependencies point to higher
// workgroup ids only and nobody waits before having done all its own work, so the launch cannot deadlock however the
// dispatcher places the workgroups; every wait is bounded all the same (error word set, the tile is then filled with NaNs:
// a timeout under a debugger or profiler turns the step's loss into NaN instead of feeding stale memory to the next layer).
struct IgemmSkArgs {
    IgemmArgs base;
    float *ws;              // [units][32][512] float4: a workgroup's accumulators of its first segment
    uint32_t *flags;        // [units + 1]: published marker per workgroup (zeroed before every launch) + an error word
    int units, tiles_total;
};
constexpr int kSkParts = 4;        // most shares a cut tile can have besides its owner's: the launcher keeps a workgroup's
                                   // run of steps >= a third of a tile
using f32x4v = decltype(__builtin_amdgcn_raw_buffer_load_b128(rsrc_t(), 0, 0, 0));

__global__ __launch_bounds__(512, 2) void conv_igemm_sk_kernel(IgemmSkArgs k) {
    using C = ICfg<64, 2>;
    const IgemmArgs &a = k.base;
    extern __shared__ __attribute__((aligned(16))) unsigned char ig_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kgrp = lane >> 5;
    const int wn = wv >> 2, wm = wv & 3;
    int unit;
    {
        const int total = (int)gridDim.x, id = (int)blockIdx.x;
        const int q = total >> 3, r = total & 7, xcd = id & 7, kk = id >> 3;
        unit = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kk;       // neighbours in the step order share an XCD
    }
    const int taps = a.taps, Cin = a.Cin, W = a.W, H = a.H;
    const int ktot = taps * Cin;
    const int nsteps = (Cin >> 6) * taps;
    const long long total_steps = (long long)k.tiles_total * nsteps;
    auto unit_start = [&](int u) { return (long long)u * total_steps / k.units; };
    const long long g1 = unit_start(unit + 1);
    const int sw = C::swz(l31);
    const uint32_t rowoff = (uint32_t)l31 * C::ROW;
    uint32_t choff[C::KS];
#pragma unroll
    for (int ks = 0; ks < C::KS; ks++) choff[ks] = rowoff + (uint32_t)(((ks * 2 + kgrp) ^ sw) << 4);
    const rsrc_t rws = make_rsrc(k.ws, (size_t)k.units * 32 * 512 * 16);
    const float floor_ = a.relu ? 0.0f : -__builtin_inff();

    for (long long g = unit_start(unit); g < g1;) {
        const int tile = (int)(g / nsteps);
        const int sb = (int)(g - (long long)tile * nsteps);
        const int se = (int)min((long long)nsteps, sb + (g1 - g));
        int t = tile;
        const int grp = t / a.tiles_per_group;
        t -= grp * a.tiles_per_group;
        const int tm = t / a.tiles_n, tn = t - tm * a.tiles_n;
        const int m0 = tm * kBM, n0 = tn * kBN;
        const IgemmGroup G = a.g[grp];
        const rsrc_t rx = make_rsrc(G.x, (size_t)a.M * Cin * 2);
        const rsrc_t rw = make_rsrc(G.w, (size_t)a.Cout * ktot * 2);
        const bool active = n0 + wn * 128 < a.Cout;

        // DMA geometry, kept in few registers (this kernel's budget is tight): instruction i of a wave moves rows wv*32 + i*8 +
        // lane/8; row and source chunk of i follow from those of i = 0 — the pixel / weight row advances by 8 rows per i, and
        // the swizzle term ((r >> 1) & 7) flips bit 2 of the chunk for odd i, i.e. XORs the byte offset with 64
        uint32_t pbase0, wbase0, pvalid01 = 0, pvalid23 = 0;
        {
            const int r0 = wv * 32 + lane / C::CPR;
            const int c0 = (lane % C::CPR) ^ C::swz(r0);
            pbase0 = (uint32_t)(m0 + r0) * (uint32_t)(Cin * 2) + (uint32_t)c0 * 16u;       // rows past M are never valid: any offset does
            wbase0 = (uint32_t)(n0 + r0) * (uint32_t)(ktot * 2) + (uint32_t)c0 * 16u;     // rows past Cout lie beyond the descriptor
#pragma unroll
            for (int i = 0; i < C::IPW; i++) {
                const int m = m0 + r0 + i * C::RPI;
                uint32_t valid = 0;
                if (m < a.M) {
                    const int hw = H * W;
                    const int bimg = m / hw, rem = m - bimg * hw, y = rem / W, x = rem - y * W;
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
                if (i < 2) pvalid01 |= valid << (16 * i); else pvalid23 |= valid << (16 * (i - 2));
            }
        }
        auto issue = [&](int stage, int s) {
            const int cc = s / taps, tap = s - cc * taps;
            int dy = 0, dx = 0;
            if (taps == 9) { dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1; }
            const int toff = (dy * G.dil * W + dx * G.dil) * Cin * 2 + cc * 128;
            unsigned char *P = ig_lds + stage * C::STAGE + wv * (32 * C::ROW);
            unsigned char *Wt = P + kBM * C::ROW;
#pragma unroll
            for (int i = 0; i < C::IPW; i++) {
                const uint32_t bits = (i < 2 ? pvalid01 : pvalid23) >> (16 * (i & 1) + tap);
                const uint32_t off = ((pbase0 + (uint32_t)(i * C::RPI * Cin * 2)) ^ ((i & 1) ? 64u : 0u)) + (uint32_t)toff;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void *)(P + i * (C::RPI * C::ROW)), 16, (bits & 1u) ? off : kOob, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < C::IPW; i++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)(Wt + i * (C::RPI * C::ROW)), 16, (i & 1) ? wbase0 ^ 64u : wbase0,
                                                         (uint32_t)s * C::ROW + (uint32_t)(i * C::RPI * ktot * 2), 0, 0);
        };

        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

        issue(0, sb);
        for (int s = sb; s < se; s++) {
            const int stage = (s - sb) & 1;
            wait_vm_barrier<0>();
            const bool more = s + 1 < se, late = wv >= 4 && active;
            if (more && !late) issue(stage ^ 1, s + 1);
            if (active) {
                const unsigned char *P = ig_lds + stage * C::STAGE + wm * (64 * C::ROW);
                const unsigned char *Wt = ig_lds + stage * C::STAGE + kBM * C::ROW + wn * (128 * C::ROW);
                // fragments of k-slice ks + 1 are read before the MFMAs of slice ks (order pinned by sched_barrier)
                bf16x8 af[2][4], bfr[2][2];
#pragma unroll
                for (int i = 0; i < 4; i++) af[0][i] = *reinterpret_cast<const bf16x8 *>(Wt + i * (32 * C::ROW) + choff[0]);
#pragma unroll
                for (int j = 0; j < 2; j++) bfr[0][j] = *reinterpret_cast<const bf16x8 *>(P + j * (32 * C::ROW) + choff[0]);
#pragma unroll
                for (int ks = 0; ks < C::KS; ks++) {
                    if (ks + 1 < C::KS) {
#pragma unroll
                        for (int i = 0; i < 4; i++) af[(ks + 1) & 1][i] = *reinterpret_cast<const bf16x8 *>(Wt + i * (32 * C::ROW) + choff[(ks + 1) % C::KS]);
#pragma unroll
                        for (int j = 0; j < 2; j++) bfr[(ks + 1) & 1][j] = *reinterpret_cast<const bf16x8 *>(P + j * (32 * C::ROW) + choff[(ks + 1) % C::KS]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int j = 0; j < 2; j++)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][i], bfr[ks & 1][j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks == 0 && more && late) issue(stage ^ 1, s + 1);
                }
            }
        }
        __syncthreads();                                    // every wave is done reading the last stage

        if (sb != 0) {
            // ---- my first segment continues a tile an earlier workgroup starts: publish the accumulators (write-through), flag
            if (active) {
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            f32x4v v = {__float_as_uint(acc[i][j][q * 4 + 0]), __float_as_uint(acc[i][j][q * 4 + 1]),
                                        __float_as_uint(acc[i][j][q * 4 + 2]), __float_as_uint(acc[i][j][q * 4 + 3])};
                            __builtin_amdgcn_raw_buffer_store_b128(v, rws, (uint32_t)(((i * 2 + j) * 4 + q) * 512 + tid) * 16u,
                                                                   (uint32_t)unit * (32u * 512u * 16u), 16);      // aux 16 = sc1
                        }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // every storing wave drains
            __syncthreads();
            if (tid == 0) __hip_atomic_store(k.flags + unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            bool ok = true;
            int nparts = 0;
            if (se != nsteps) {
                // ---- I hold the tile's first steps: the rest are the first segments of the following workgroups.  Wait for
                // their flags (one lane, relaxed polls, bounded), ONE acquire, barrier; the epilogue below then adds their
                // accumulators group by group, in workgroup order.
                const long long tile_end = (long long)(tile + 1) * nsteps;
                while (unit + 1 + nparts < k.units && unit_start(unit + 1 + nparts) < tile_end) nparts++;
                volatile int &sk_ok = *reinterpret_cast<volatile int *>(ig_lds);     // the stages are idle here (one LDS object only)
                if (tid == 0) {
                    int good = nparts <= kSkParts ? 1 : 0;          // (the launcher's rule makes more impossible; never drop a share silently)
                    for (int p = 0; p < nparts && good; p++) {
                        unsigned spins = 0;
                        while (__hip_atomic_load(k.flags + unit + 1 + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) {
                            __builtin_amdgcn_s_sleep(8);
                            if (++spins > (1u << 24)) { good = 0; break; }           // ~ seconds: give up, report
                        }
                    }
                    if (good) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    else __hip_atomic_store(k.flags + k.units, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sk_ok = good;
                }
                __syncthreads();
                ok = sk_ok != 0;
                __syncthreads();                              // the flag word is an epilogue row again
            }
            if (active) {
                // (a share that never arrived within the bounded wait — error word set — poisons the tile: every value gets a NaN
                // added, so the step's loss is NaN instead of stale memory feeding the next layer silently)
                const float poison = ok ? 0.0f : __builtin_nanf("");
                // ---- epilogue of the complete tile (as conv_igemm_body), the other workgroups' shares of a cut tile added group
                // by group in workgroup order: the up to kSkParts 16-byte loads of group gi + 1 are in flight while group gi is
                // finished (a share beyond nparts gets an out-of-range offset = zeros, so the code has no branch per share)
                unsigned char *O = ig_lds + wv * kOutWave;
                const int nw = n0 + wn * 128;
                const rsrc_t rb = make_rsrc(G.bias, G.bias ? (size_t)a.Cout * 4 : 0);
                const uint32_t seed_g = a.seed_lo + (uint32_t)grp * 0x9E3779B9u;
                f32x4v shares[2][kSkParts];
                float bq[2][4];
                auto fetch = [&](int gi, f32x4v (&dst)[kSkParts], float (&bias4)[4]) {
                    const int i = gi >> 3, q = (gi >> 1) & 3, j = gi & 1;
                    const uint32_t off = (uint32_t)(((i * 2 + j) * 4 + q) * 512 + tid) * 16u;
#pragma unroll
                    for (int p = 0; p < kSkParts; p++)
                        dst[p] = __builtin_amdgcn_raw_buffer_load_b128(rws, p < nparts ? off : kOob, (uint32_t)(unit + 1 + p) * (32u * 512u * 16u), 0);
#pragma unroll
                    for (int e = 0; e < 4; e++) bias4[e] = ld_f32(rb, (uint32_t)(nw + i * 32 + q * 8 + kgrp * 4 + e) * 4u);
                };
                fetch(0, shares[0], bq[0]);
#pragma unroll
                for (int gi = 0; gi < 32; gi++) {
                    const int i = gi >> 3, q = (gi >> 1) & 3, j = gi & 1;
                    if (gi + 1 < 32) fetch(gi + 1, shares[(gi + 1) & 1], bq[(gi + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    const int nl = i * 32 + q * 8 + kgrp * 4;
                    float s0 = acc[i][j][q * 4 + 0], s1 = acc[i][j][q * 4 + 1], s2 = acc[i][j][q * 4 + 2], s3 = acc[i][j][q * 4 + 3];
#pragma unroll
                    for (int p = 0; p < kSkParts; p++) {
                        const f32x4v pv = shares[gi & 1][p];
                        s0 += __uint_as_float(pv[0]); s1 += __uint_as_float(pv[1]); s2 += __uint_as_float(pv[2]); s3 += __uint_as_float(pv[3]);
                    }
                    float v0 = s0 + bq[gi & 1][0], v1 = s1 + bq[gi & 1][1], v2 = s2 + bq[gi & 1][2], v3 = s3 + bq[gi & 1][3];
                    v0 = fmaxf(v0, floor_) + poison; v1 = fmaxf(v1, floor_) + poison; v2 = fmaxf(v2, floor_) + poison; v3 = fmaxf(v3, floor_) + poison;
                    if (a.drop_thresh) {
                        const uint32_t m = (uint32_t)(m0 + wm * 64 + j * 32 + l31);
                        const uint32_t h = dropout_bytes((m * (uint32_t)a.Cout + (uint32_t)(nw + nl)) >> 2, seed_g, a.seed_hi);
                        v0 = (h & 0xffu) >= a.drop_thresh ? v0 * a.drop_scale : 0.0f;
                        v1 = ((h >> 8) & 0xffu) >= a.drop_thresh ? v1 * a.drop_scale : 0.0f;
                        v2 = ((h >> 16) & 0xffu) >= a.drop_thresh ? v2 * a.drop_scale : 0.0f;
                        v3 = (h >> 24) >= a.drop_thresh ? v3 * a.drop_scale : 0.0f;
                    }
                    *reinterpret_cast<uint2 *>(O + (j * 32 + l31) * kOutRow + nl * 2) = make_uint2(pack2(v0, v1), pack2(v2, v3));
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll 4
                for (int it = 0; it < 16; it++) {
                    const int p = it * 4 + (lane >> 4), ch = lane & 15;
                    const int m = m0 + wm * 64 + p;
                    const uint4 v = *reinterpret_cast<const uint4 *>(O + p * kOutRow + ch * 16);
                    if (m < a.M) *reinterpret_cast<uint4 *>(G.y + (size_t)m * a.Cout + nw + ch * 8) = v;
                }
            }
        }
        g += se - sb;
        __syncthreads();                                    // the epilogue's LDS rows are the next segment's stages
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same convolutions, again without an im2col matrix:
//
//   gw[n][tap][c] = sum over pixels m of  g[m][n] * x[pixel(m) + tap offset][c]
//
// A GEMM whose reduction runs over the pixels — the SLOW axis of both NHWC operands — so both MFMA operands are formed by
// gfx950's transposing LDS read (ds_read_b64_tr_b16: a 16-lane group reads a [4 pixels][16 channels] block and each lane
// receives one channel's four pixels), exactly as the direct kernels' weight gradient does (conv_direct.hip), but with the
// tiles DMA'd global -> LDS as in the forward kernel above:
//   * output tile 256 channels of g x 256 columns (one tap, 256 channels of x), K-step = 64 pixels: two tiles of 64 rows x
//     512 bytes per step; a DMA wave instruction moves 2 rows; the 64-byte piece (32 channels) p of row r is stored at piece
//     position p ^ (r & 3) so that the four pixel rows of a transposing read hit four different bank quarters;
//   * the x rows are the g rows' pixels shifted by the tile's tap (zero outside the map, per lane, by the descriptor's range
//     check); (row, column) of a lane's pixels advance incrementally by 64 pixels per step — no division in the loop;
//   * 9 * cin * cout / 65536 tiles are far fewer than CUs, so the pixel range is split over `ksplit` workgroups per tile
//     (chosen by the launcher to fill whole rounds of the chip); each writes its fp32 partial tile, and a second kernel sums
//     the partials in a fixed order (deterministic) into the gradient, laid out [n][tap][c] = a channels_last (cout, cin, k, k)
//     tensor, in fp32 (the master weights' gradient needs no further cast) or bf16.
struct WgradGroup {
    const uint16_t *x;      // (B, H, W, Cin) bf16: the layer's input
    const uint16_t *g;      // (B, H, W, Cout) bf16: gradient of its output
    float *part;            // (ksplit, Cout, taps, Cin) f32 partials
    int dil, pad_;
};
struct IgemmWgradArgs {
    WgradGroup g[4];
    int ngroups, B, H, W, Cin, Cout, taps, M, tiles_n, tiles_c, ksplit, kchunk, tiles_per_group;
    int stagger;            // as IgemmArgs::stagger
    int compact;            // 1: a tile sums over the pixels its tap reaches only (see the kernel); every split takes an equal share of THEM
    int xcd_mix;            // 1: XCD-interleaved map of the (group, pixel chunk, tile) workgroups (see the kernel); needs 8 | ksplit
    int skip_rows;          // 1: a K-step (64 pixels) whose rows the tile's tap shifts out of the map entirely is not loaded and
                            // not multiplied (its x rows would all be padding zeros)
};
constexpr int kWRow = 512;                                 // bytes per LDS row: 256 channels
constexpr int kWTile = 64 * kWRow;                         // 64 pixels
constexpr int kWStage = 2 * kWTile;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4v;
typedef __attribute__((address_space(3))) unsigned char lds_u8;

__device__ __forceinline__ bf16x8 tr_frag8(lds_u8 *p) {    // pixels 0-3 and 4-7 of a lane's k-range: two transposing reads
    bf16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(reinterpret_cast<__attribute__((address_space(3))) bf16x4v *>(p));
    bf16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(reinterpret_cast<__attribute__((address_space(3))) bf16x4v *>(p + 4 * kWRow));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ void conv_igemm_wgrad_body(const IgemmWgradArgs &a, const int bid, const int nblk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ig_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kgrp = lane >> 5, i16 = lane & 15, gq = lane >> 4;
    const int wn = wv >> 2, wm = wv & 3;

    // all tiles of one pixel chunk are neighbours in t (they read the same rows of g and x): they share an XCD's L2.
    // xcd_mix (launches that skip rows; 8 | ksplit): workgroups no longer cost the same (a dilation-24 tile skips 37 % of its
    // steps, a dilation-6 one 7 %) and an XCD only runs the blocks with its id % 8 — XCD x takes the pixel chunks x, x + 8, ..
    // of EVERY group and tile (order: group, chunk, tile), which keeps a chunk's tiles on one XCD and gives all XCDs the same mix.
    const int tiles = a.tiles_n * a.tiles_c;
    int grp, split, t;
    if (a.xcd_mix) {
        const int id = bid, xcd = id & 7, k = id >> 3;
        if (a.ksplit >= 8) {                                // 8 | ksplit: XCD x owns the chunks x, x + 8, ..
            const int per_group = (a.ksplit >> 3) * tiles;
            grp = k / per_group;
            const int r = k - grp * per_group, sl = r / tiles;
            t = r - sl * tiles;
            split = xcd + 8 * sl;
        } else {                                            // ksplit | 8: a chunk belongs to 8 / ksplit XCDs, its tiles dealt out among them in turn
            const int xpc = 8 / a.ksplit, sub = xcd % xpc, per_group = tiles / xpc;      // (xpc | tiles: the launcher checked)
            split = xcd / xpc;
            grp = k / per_group;
            t = sub + xpc * (k - grp * per_group);
        }
    } else {
        const int total = nblk, id = bid;
        const int q = total >> 3, r = total & 7, xcd = id & 7, k = id >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
        grp = t / a.tiles_per_group;
        t -= grp * a.tiles_per_group;
        split = t / tiles;
        t -= split * tiles;
    }
    const int tn = t / a.tiles_c, tc = t - tn * a.tiles_c;
    const WgradGroup G = a.g[grp];
    const int taps = a.taps, Cin = a.Cin, Cout = a.Cout, W = a.W, H = a.H;
    // the 256 columns of a tile: one tap x 256 channels of x — or, for a 128-channel x (conv3_1), two taps x 128 channels
    const bool two = Cin == 128 && taps == 9;
    const int ncb = two ? 1 : (Cin + 255) >> 8;             // 256-channel blocks of x per tap (the last one may be narrower)
    const int tap = two ? 2 * tc : tc / ncb, c0 = two ? 0 : (tc - tap * ncb) << 8, n0 = tn << 8;
    // ---- the pixels this tile sums over.  The reduction index q runs over the pixels of a RECTANGLE of every image, image-major,
    // raster order inside: the whole map — or (compact: dilated kernels) exactly the pixels whose tap lands inside the map: a tap of
    // dilation 24 on a 41 x 41 map reaches 17 x 17 (corner tap) or 17 x 41 (edge tap) pixels, everything else would multiply padding
    // zeros.  The K loop then holds no dead element at all (36 % of the four fc6_k's pixel x tap pairs are dead; skipping whole
    // 64-pixel steps of the flat order catches 22 %), and the terms that remain are summed in the order they had before.  Every
    // 3x3 layer takes this form: with dilation 1 / 2 the rectangle only loses the one or two border rows and columns (2-10 % of
    // the pixels), and every split of a tile gets an equal share of the LIVE pixels.
    const int tdy = (taps == 9 && !two) ? (tap / 3 - 1) * G.dil : 0, tdx = (taps == 9 && !two) ? (tap % 3 - 1) * G.dil : 0;
    const bool compact = a.compact && !two && taps == 9;
    const int ry0 = compact ? max(0, -tdy) : 0, rx0 = compact ? max(0, -tdx) : 0;
    const int rh_ = compact ? H - abs(tdy) : H, rw_ = compact ? W - abs(tdx) : W;
    const int rh = max(rh_, 1), rw = max(rw_, 1);                         // (an empty rectangle: no steps at all, see Kc)
    const int area = rh * rw;
    const int Kc = (rh_ > 0 && rw_ > 0) ? a.B * area : 0;
    const int kchunk = compact ? (((Kc + a.ksplit - 1) / a.ksplit + 63) >> 6) << 6 : a.kchunk;
    const int mbeg = split * kchunk, mend = min(Kc, mbeg + kchunk);      // [mbeg, mend) of the reduction index
    const int nsteps = mend > mbeg ? (mend - mbeg + 63) >> 6 : 0;

    const rsrc_t rx = make_rsrc(G.x, (size_t)a.M * Cin * 2);
    const rsrc_t rg = make_rsrc(G.g, (size_t)a.M * Cout * 2);

    // ---- DMA geometry: per step a wave moves rows [wv*8 + i*2, +2) of both tiles, i = 0..3; lane -> (row, 16-byte piece)
    int pm[4], pb[4], py[4], px[4], ldy[4], ldx[4], ltapoff[4];       // reduction index, image, row and column inside the rectangle
    uint32_t srcoff[4], xsrcoff[4];                         // byte offset of the lane's source piece inside a row of g / of x
    bool ltap[4];                                           // the lane's tap exists (a two-tap tile may hold the tenth)
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = wv * 8 + i * 2 + (lane >> 5);
        const int q = lane & 31, p = (q >> 2) ^ (r & 3);    // 64-byte source piece that lands at piece position q >> 2
        srcoff[i] = (uint32_t)(p * 64 + (q & 3) * 16);
        const int mytap = two ? tap + (p >> 2) : tap;       // pieces 4-7 of a two-tap tile belong to the second tap
        xsrcoff[i] = two ? (uint32_t)((p & 3) * 64 + (q & 3) * 16) : srcoff[i];
        // channel counts that are no multiple of 256 (ResNet res2 / res3: 64, 128): the pieces of a 512-byte tile row past the tensor's
        // own row must read zeros — bit 31 of the piece offset pushes the lane's address beyond the descriptor (both tensors < 2^31 bytes)
        if (!two && (uint32_t)(c0 * 2) + xsrcoff[i] >= (uint32_t)(Cin * 2)) xsrcoff[i] |= 0x80000000u;
        if ((uint32_t)(n0 * 2) + srcoff[i] >= (uint32_t)(Cout * 2)) srcoff[i] |= 0x80000000u;
        ltap[i] = mytap < taps;
        ldy[i] = taps == 9 ? (mytap / 3 - 1) * G.dil : 0;
        ldx[i] = taps == 9 ? (mytap % 3 - 1) * G.dil : 0;
        ltapoff[i] = (ldy[i] * W + ldx[i]) * Cin * 2 + c0 * 2;
        const int m = mbeg + r;
        pm[i] = m;
        const int mm = m < Kc ? m : 0;
        pb[i] = mm / area;
        const int rem = mm - pb[i] * area;
        py[i] = rem / rw;
        px[i] = rem - py[i] * rw;
    }
    const int qW = 64 / rw, rW = 64 - qW * rw;              // a step advances every row by 64 elements of the reduction index

    auto advance = [&]() {                                   // the lanes' pixels move on by one step
#pragma unroll
        for (int i = 0; i < 4; i++) {
            pm[i] += 64;
            px[i] += rW;
            py[i] += qW;
            if (px[i] >= rw) { px[i] -= rw; py[i] += 1; }
            while (py[i] >= rh) { py[i] -= rh; pb[i] += 1; }
        }
    };
    auto issue = [&](int stage) {
        unsigned char *A = ig_lds + stage * kWStage + wv * (8 * kWRow);
        unsigned char *Bt = A + kWTile;
        int pix[4];
#pragma unroll
        for (int i = 0; i < 4; i++) pix[i] = compact ? (pb[i] * H + ry0 + py[i]) * W + rx0 + px[i] : pm[i];       // the pixel of the map
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool live = pm[i] < mend;
            const uint32_t vo = live ? (uint32_t)pix[i] * (uint32_t)(Cout * 2) + (uint32_t)(n0 * 2) + srcoff[i] : kOob;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lds_void *)(A + i * (2 * kWRow)), 16, vo, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int yy = ry0 + py[i] + ldy[i], xx = rx0 + px[i] + ldx[i];
            const bool live = ltap[i] && pm[i] < mend && yy >= 0 && yy < H && xx >= 0 && xx < W;
            const uint32_t vo = live ? (uint32_t)(pix[i] * (Cin * 2) + ltapoff[i]) + xsrcoff[i] : kOob;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void *)(Bt + i * (2 * kWRow)), 16, vo, 0, 0, 0);
        }
        advance();                                           // the next step's pixels
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    // the address a lane supplies to a transposing read of fragment f (32 channels = one 64-byte piece), k-slice ks: pixel row
    // ks*16 + kgrp*8 + i16/4 (the second read 4 rows further), piece f ^ (row & 3), channels (gq & 1)*16 + 4 (i16 & 3) .. + 3
    const int swz = (i16 >> 2) & 3;
    const uint32_t lane_off = (uint32_t)((kgrp * 8 + (i16 >> 2)) * kWRow + (gq & 1) * 32 + (i16 & 3) * 8);
    uint32_t aoff[4], boff[2];
#pragma unroll
    for (int i = 0; i < 4; i++) aoff[i] = lane_off + (uint32_t)(((wn * 4 + i) ^ swz) * 64);
#pragma unroll
    for (int j = 0; j < 2; j++) boff[j] = lane_off + (uint32_t)(kWTile + ((wm * 2 + j) ^ swz) * 64);
    lds_u8 *lds = (lds_u8 *)ig_lds;

    // ---- the K-steps this tile multiplies.  A dilated tap shifts whole rows of the map into the padding: with dilation 24 on 41
    // rows the three upper taps see x only from rows >= 24 — for the 64-pixel steps that lie in rows 0..23 of an image every
    // x row is zero and the step adds exact zeros (22 % of the four fc6_k's steps at 16 images, 37 % for dilation 24).  Such
    // steps are skipped: the lanes' pixel counters run on, nothing is loaded, nothing multiplied; the partial sums are
    // bit-identical.  Decided per step from the rows its first and last pixel lie in (wave-uniform, scalar): a step inside one
    // image is dead when no row in [y_first, y_last] lands in [0, H) after the shift; a step that crosses into the next image
    // always holds live rows of one of the two (|shift| < H), and so does any tile without a vertical shift.
    // (a step of 64 pixels spans at least 64 / W rows: below that shift no step can be dead, and the test is not free)
    const bool may_skip = a.skip_rows && !compact && abs(tdy) * W >= 64 && tdy > -H && tdy < H;
    const int hw_ = H * W;
    auto step_dead = [&](int st) -> bool {
        const int p0 = mbeg + st * 64, p1 = min(p0 + 63, mend - 1);
        const int b0 = p0 / hw_, b1 = p1 / hw_;
        if (b0 != b1) return false;
        const int y0 = (p0 - b0 * hw_) / W, y1 = (p1 - b0 * hw_) / W;
        return y1 + tdy < 0 || y0 + tdy >= H;
    };
    int cur = 0;                                             // the step the lanes' counters stand at = the next one to issue
    auto seek = [&]() {
        if (may_skip)
            while (cur < nsteps && step_dead(cur)) { advance(); cur++; }
    };
    seek();
    bool have = cur < nsteps;
    if (have) { issue(0); cur++; }
    for (int s = 0; have; s++) {
        wait_vm_barrier<0>();
        seek();
        const bool more = cur < nsteps, late = a.stagger && wv >= 4;
        if (more && !late) { issue((s + 1) & 1); cur++; }
        lds_u8 *st = lds + (s & 1) * kWStage;
        bf16x8 af[2][4], bfr[2][2];
#pragma unroll
        for (int i = 0; i < 4; i++) af[0][i] = tr_frag8(st + aoff[i]);
#pragma unroll
        for (int j = 0; j < 2; j++) bfr[0][j] = tr_frag8(st + boff[j]);
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            if (ks + 1 < 4) {
#pragma unroll
                for (int i = 0; i < 4; i++) af[(ks + 1) & 1][i] = tr_frag8(st + (ks + 1) * (16 * kWRow) + aoff[i]);
#pragma unroll
                for (int j = 0; j < 2; j++) bfr[(ks + 1) & 1][j] = tr_frag8(st + (ks + 1) * (16 * kWRow) + boff[j]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][i], bfr[ks & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 0 && more && late) { issue((s + 1) & 1); cur++; }
        }
        have = more;
    }
    // C[row = channel of g][col = channel of x]: lane holds column l31, rows (reg & 3) + 8 (reg >> 2) + 4 kgrp
    float *pp = G.part + (size_t)split * ((size_t)Cout * taps * Cin);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int col = wm * 64 + j * 32 + l31;
            const int etap = two ? tap + (col >> 7) : tap, c = two ? (col & 127) : c0 + col;
            if (etap < taps && c < Cin && n0 + wn * 128 + i * 32 < Cout) {      // (32-row / 32-column blocks: channel counts are multiples of 64)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int n = n0 + wn * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kgrp;
                    pp[((size_t)n * taps + etap) * Cin + c] = acc[i][j][r];
                }
            }
        }
}

__global__ __launch_bounds__(512, 2) void conv_igemm_wgrad_kernel(IgemmWgradArgs a) { conv_igemm_wgrad_body(a, (int)blockIdx.x, (int)gridDim.x); }

// One launch for the whole backward of a layer's convolution: both gradients need nothing but the (masked) output gradient, and a
// 41x41 layer's data gradient is 212 tiles for 256 CUs — one round with a sixth of the chip idle — while its weight gradient is
// 252 workgroups.  Blocks [0, nd) are the data gradient's tiles, blocks [nd_pad, nd_pad + nw) the weight gradient's workgroups
// (nd_pad = nd rounded up to a multiple of 8, so that both halves keep their block-id -> XCD relation; the padding blocks exit):
// the dispatcher hands the weight gradient's first workgroups to the CUs the data gradient leaves idle, no stream, no join.
// (The two kernels on two streams lost: 1 761 -> 1 715 images/s, profiles/r05_wgrad_side_stream_ab.txt.)
struct IgemmBwdArgs {
    IgemmArgs d;
    IgemmWgradArgs w;
    int nd, nd_pad, nw;
    int w_first, nw_pad;    // 1: the weight gradient's workgroups take the first block ids (they are dispatched first): the order for a layer
                            // whose weight-gradient workgroups run longer than its data-gradient tiles (a 1x1 layer: tiles of 4 - 16 K-steps)
};
__global__ __launch_bounds__(512, 2) void conv_igemm_bwd_kernel(IgemmBwdArgs a) {
    const int id = (int)blockIdx.x;
    if (a.w_first) {
        if (id < a.nw_pad) {
            if (id < a.nw) conv_igemm_wgrad_body(a.w, id, a.nw);
        } else if (id - a.nw_pad < a.nd) {
            conv_igemm_body<64, 2>(a.d, id - a.nw_pad, a.nd);
        }
        return;
    }
    if (id < a.nd_pad) {
        if (id < a.nd) conv_igemm_body<64, 2>(a.d, id, a.nd);
    } else {
        conv_igemm_wgrad_body(a.w, id - a.nd_pad, a.nw);
    }
}

// gw[e] = sum over the splits of part[s][e], e over [n][tap][c], in split order; float4 per thread; blockIdx.y = the group
// (the four branches of a grouped launch share ONE reduction launch: 17 -> 11 reduction launches per train step)
struct WgradReduceArgs {
    const float *part[4];
    void *gw[4];
    int ksplit;
    size_t n4;
    const float *scale;     // nullptr, or (cout) f32: output channel o's gradient times scale[o] (a constant per-channel factor folded into
    size_t per_o4;          // the kernel the convolution ran with: d/dw = scale * d/d(w scale)); per_o4 = float4s per output channel
};
template <bool BF16_OUT>
__global__ __launch_bounds__(256) void conv_igemm_wgrad_reduce_kernel(WgradReduceArgs a) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= a.n4) return;
    const float4 *p = reinterpret_cast<const float4 *>(a.part[blockIdx.y]) + e;
    void *gw = a.gw[blockIdx.y];
    float4 s = p[0];
    int k = 1;
    for (; k + 8 <= a.ksplit; k += 8) {                      // eight loads in flight, added in split order (the sum's bits do not change)
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = p[(size_t)(k + u) * a.n4];
#pragma unroll
        for (int u = 0; u < 8; u++) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; k < a.ksplit; k++) {
        const float4 v = p[(size_t)k * a.n4];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (a.scale) {
        const float f = a.scale[e / a.per_o4];
        s.x *= f; s.y *= f; s.z *= f; s.w *= f;
    }
    if (BF16_OUT) reinterpret_cast<uint2 *>(gw)[e] = make_uint2(pack2(s.x, s.y), pack2(s.z, s.w));
    else reinterpret_cast<float4 *>(gw)[e] = s;
}


// ------------------------------------------------------------------------------------------------------------------
// Both packed forms of a kernel in one pass over the float32 master weights ([o][tap][c] = a channels_last (cout, cin, k, k)
// parameter): the forward form fwd[o][c / 64][tap][64] and the data-gradient form dg[c][o / 64][T - tap][64] (the kernel
// flipped, T = k*k - 1, and its channel axes swapped), both bf16.  One block per (64 outputs, 64 inputs, tap): the forward
// form is the tile as it is read; the data-gradient form is its transpose, taken through LDS.
// plain = 1: the layouts of the direct kernels (conv_direct.hip) instead — fwd[o][tap][c] (the parameter's own order, cast) and
// dg[c][T - tap][o] (a channels_last (cin, cout, k, k) tensor: the kernel flipped, channel axes swapped).
__global__ __launch_bounds__(256) void pack_conv_weight_kernel(const float *w, uint16_t *fwd, uint16_t *dg, int cout, int cin, int taps,
                                                               int plain, const float *scale) {      // scale: nullptr or (cout) — w[o] * scale[o] is packed
    __shared__ uint16_t tile[64][64 + 4];
    const int ob = blockIdx.x, cb = blockIdx.y, tap = blockIdx.z, t = threadIdx.x;
    const int r = t >> 4, q = t & 15;                        // 16 rows x 16 float4 per pass
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int o = r + 16 * i;
        const size_t src = ((size_t)(ob * 64 + o) * taps + tap) * cin + cb * 64 + q * 4;
        float4 v = *reinterpret_cast<const float4 *>(w + src);
        if (scale) {
            const float f = scale[ob * 64 + o];
            v.x *= f; v.y *= f; v.z *= f; v.w *= f;
        }
        const uint2 pk = make_uint2(pack2(v.x, v.y), pack2(v.z, v.w));
        if (fwd) *reinterpret_cast<uint2 *>(fwd + (plain ? src : (((size_t)(ob * 64 + o) * (cin >> 6) + cb) * taps + tap) * 64 + q * 4)) = pk;
        *reinterpret_cast<uint2 *>(&tile[o][q * 4]) = pk;
    }
    if (!dg) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c = r + 16 * i;                            // row of the transposed tile; q*4 .. q*4+3 = its outputs
        const uint32_t lo = (uint32_t)tile[q * 4 + 0][c] | ((uint32_t)tile[q * 4 + 1][c] << 16);
        const uint32_t hi = (uint32_t)tile[q * 4 + 2][c] | ((uint32_t)tile[q * 4 + 3][c] << 16);
        const size_t dst = plain ? ((size_t)(cb * 64 + c) * taps + (taps - 1 - tap)) * cout + ob * 64 + q * 4
                                 : (((size_t)(cb * 64 + c) * (cout >> 6) + ob) * taps + (taps - 1 - tap)) * 64 + q * 4;
        *reinterpret_cast<uint2 *>(dg + dst) = make_uint2(lo, hi);
    }
}

}  // namespace

bool conv_igemm_supported(int cin, int cout, int k) {
    return (k == 1 || k == 3) && cin >= 64 && cin % 64 == 0 && cout >= 128 && cout % 128 == 0;
}
// what the launch takes: also a 64-channel output (a wave's 128 output columns half empty: rows of w past cout read zeros through the
// descriptor, the store skips them) — not the recommended route for a 3x3 layer (conv_direct.hip), but a 1x1 layer over many pixels
// is bandwidth-bound either way (ResNet res2: 256 -> 64 at 129 x 129 x 10 pixels)
static bool conv_igemm_launchable(int cin, int cout, int k) {
    return (k == 1 || k == 3) && cin >= 64 && cin % 64 == 0 && (cout == 64 || (cout >= 128 && cout % 128 == 0));
}

std::atomic<int> g_igemm_variant{-1};      // dsrg_debug_set_igemm_variant (tests / tools); -1 = DSRG_IGEMM_VARIANT or the default
static int igemm_variant() {
    int v = g_igemm_variant.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("DSRG_IGEMM_VARIANT");
        v = e ? atoi(e) : 3;
        g_igemm_variant.store(v, std::memory_order_relaxed);
    }
    return v;
}

static int igemm_cus() {
    static const int n = [] {
        int dev = 0, c = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c < 8) c = 256;
        return c;
    }();
    return n;
}
// device scratch of the stream-K form: one accumulator tile (256 KB) and one flag per workgroup (= per CU), an error word
size_t conv_igemm_workspace() { return (size_t)igemm_cus() * (32 * 512 * 16) + ((size_t)igemm_cus() + 1 + 63) / 64 * 256; }

// bias gradient from the per-tile column sums of a masked data gradient: fixed order (sixteen interleaved row streams, then
// a tree), one workgroup per 64 channels of a group
struct ColsumArgs {
    const float *part[4];
    float *out[4];
    int rows, cout;
};
__global__ __launch_bounds__(1024) void igemm_colsum_kernel(ColsumArgs a) {
    __shared__ float red[16][64];
    colsum_block(a.part[blockIdx.y], a.out[blockIdx.y], a.rows, a.cout, (int)blockIdx.x, red);
}

int launch_igemm_colsum(const float *const *parts, float *const *outs, int ngroups, int rows, int cout, hipStream_t stream) {
    if (ngroups < 1 || ngroups > 4 || cout % 64 || rows < 1) return set_error(DSRG_ERR_INVALID, "column sums: 1..4 groups, 64 | channels");
    {   // deferred (dsrg_defer_reductions): all groups or none
        int g = 0;
        while (g < ngroups && defer_reduction(0, parts[g], outs[g], rows, cout)) g++;
        if (g == ngroups) return DSRG_OK;
        if (g > 0) return set_error(DSRG_ERR_INVALID, "column sums: the deferred-reduction list filled up in the middle of a launch's groups");
    }
    ColsumArgs c;
    memset(&c, 0, sizeof(c));
    for (int g = 0; g < ngroups; g++) { c.part[g] = parts[g]; c.out[g] = outs[g]; }
    c.rows = rows; c.cout = cout;
    hipLaunchKernelGGL(igemm_colsum_kernel, dim3(cout / 64, ngroups), dim3(1024), 0, stream, c);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

// row-aligned pixel tiles (IgemmArgs::row_tiles) when at least two rows fit a tile and the rows a band leaves empty cost less
// than a tenth of the tiles
static bool conv_igemm_row_tiles(int H, int W) {
    if (W > kBM / 2) return false;
    const int r = kBM / W, bands = (H + r - 1) / r;
    return (long long)bands * kBM * 10 <= (long long)H * W * 11 + 10LL * kBM;
}
// most pixel tiles a launch over B maps of H x W can have per group (flattened or row-aligned): sizes the column-sum scratch
static size_t conv_igemm_pixel_tiles(int B, int H, int W) {
    const long long M = (long long)B * H * W;
    size_t t = (size_t)((M + kBM - 1) / kBM);
    if (W <= kBM && conv_igemm_row_tiles(H, W)) {
        const int r = kBM / W;
        const size_t tr = (size_t)B * ((H + r - 1) / r);
        if (tr > t) t = tr;
    }
    return t;
}
size_t conv_igemm_colsum_workspace(int ngroups, int B, int H, int W, int cout) {
    return (size_t)ngroups * conv_igemm_pixel_tiles(B, H, W) * (size_t)cout * sizeof(float);
}

// ---- class order of a dilated launch's pixels (IgemmArgs::cls_tiles)
static uint32_t tap_mask_rect(int H, int W, int d, int y0, int y1, int x0, int x1) {      // the taps that reach a pixel of [y0, y1) x [x0, x1)
    uint32_t m = 0;
    for (int tap = 0; tap < 9; tap++) {
        const int dy = (tap / 3 - 1) * d, dx = (tap % 3 - 1) * d;
        if (std::max(y0, -dy) < std::min(y1, H - dy) && std::max(x0, -dx) < std::min(x1, W - dx)) m |= 1u << tap;
    }
    return m;
}
// the bands of an axis of n pixels inside each of which the taps -d / +d are either valid for every pixel or for none
static int axis_bands(int n, int d, int (*out)[2]) {
    if (d >= n) { out[0][0] = 0; out[0][1] = n; return 1; }
    const int lo = std::min(d, n - d), hi = std::max(d, n - d), cut[4] = {0, lo, hi, n};
    int k = 0;
    for (int i = 0; i < 3; i++)
        if (cut[i + 1] > cut[i]) { out[k][0] = cut[i]; out[k][1] = cut[i + 1]; k++; }
    return k;
}
struct HostClass { int y0, y1, x0, x1; uint32_t mask; };
static int build_classes(int H, int W, int d, HostClass *c) {      // most live taps first (ties: map order)
    int yb[3][2], xb[3][2];
    const int ny = axis_bands(H, d, yb), nx = axis_bands(W, d, xb);
    int n = 0;
    for (int i = 0; i < ny; i++)
        for (int j = 0; j < nx; j++) c[n++] = HostClass{yb[i][0], yb[i][1], xb[j][0], xb[j][1], tap_mask_rect(H, W, d, yb[i][0], yb[i][1], xb[j][0], xb[j][1])};
    std::stable_sort(c, c + n, [](const HostClass &a, const HostClass &b) { return __builtin_popcount(a.mask) > __builtin_popcount(b.mask); });
    return n;
}
// K-steps per 64-channel chunk (= live taps summed over the pixel tiles) of one group: class order / row-aligned / flattened tiles
static long long tile_taps(int B, int H, int W, int d, int mode) {
    const long long M = (long long)B * H * W;
    long long tot = 0;
    if (mode == 2) {                                         // class order
        HostClass c[kMaxClasses];
        const int n = build_classes(H, W, d, c);
        long long q0[kMaxClasses + 1];
        q0[0] = 0;
        for (int k = 0; k < n; k++) q0[k + 1] = q0[k] + (long long)B * (c[k].y1 - c[k].y0) * (c[k].x1 - c[k].x0);
        for (long long t0 = 0; t0 < M; t0 += kBM) {
            const long long t1 = std::min(M, t0 + kBM);
            uint32_t m = 0;
            for (int k = 0; k < n; k++)
                if (q0[k] < t1 && q0[k + 1] > t0) m |= c[k].mask;
            tot += __builtin_popcount(m);
        }
    } else if (mode == 1) {                                  // whole rows of one image
        const int r = kBM / W;
        for (int y0 = 0; y0 < H; y0 += r) tot += (long long)B * __builtin_popcount(tap_mask_rect(H, W, d, y0, std::min(H, y0 + r), 0, W));
    } else {                                                 // 256 consecutive pixels
        for (long long t0 = 0; t0 < M; t0 += kBM) {
            const long long t1 = std::min(M, t0 + kBM);
            uint32_t m = 0;
            for (long long p = t0; p < t1;) {                // row by row (a row piece is a rectangle)
                const int rem = (int)(p % ((long long)H * W)), y = rem / W, x = rem % W;
                const int x1 = (int)std::min<long long>(W, x + (t1 - p));
                m |= tap_mask_rect(H, W, d, y, y + 1, x, x1);
                p += x1 - x;
            }
            tot += __builtin_popcount(m);
        }
    }
    return tot;
}

// set by launch_conv_igemm_backward around its calls of the two launchers below: they then validate, fill the argument block
// and return it instead of launching
static thread_local IgemmArgs *t_prep_d = nullptr;
static thread_local IgemmWgradArgs *t_prep_w = nullptr;
static thread_local int *t_prep_grid = nullptr;
static thread_local int t_force_ksplit = 0;              // launch_conv_igemm_backward: the pixel split it picked for its merged grid
static thread_local const float *t_wgrad_scale = nullptr; // launch_conv_igemm_backward_residual: per-output factor of the weight gradient (nullptr: none)
static thread_local const void *t_res = nullptr;         // launch_conv_igemm_residual: the residual of its single group (nullptr: none)
static thread_local int t_split_cin = 0;                 // launch_conv_igemm_split: the real input channel count (0: ordinary launch)

int launch_conv_igemm(const void *const *x, const void *const *w, const float *const *bias, void *const *y, const int *dil,
                      int ngroups, int B, int H, int W, int cin, int cout, int k, int relu, float drop_p, unsigned long long seed,
                      void *workspace, size_t workspace_bytes, hipStream_t stream, const void *const *mask, float out_scale,
                      float *const *colsum, void *colsum_ws, size_t colsum_ws_bytes) {
    if (ngroups < 1 || ngroups > 4) return set_error(DSRG_ERR_INVALID, "conv_igemm: 1..4 groups");
    if (colsum && (!colsum_ws || colsum_ws_bytes < conv_igemm_colsum_workspace(ngroups, B, H, W, cout)))
        return set_error(DSRG_ERR_INVALID, "conv_igemm: column-sum scratch missing or too small");
    if (!conv_igemm_launchable(cin, cout, k))
        return set_error(DSRG_ERR_UNSUPPORTED, "conv_igemm: cin %% 64 == 0, cout %% 128 == 0 (or cout = 64), k in (1, 3) required (got %d, %d, %d)",
                         cin, cout, k);
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
        a.g[g].mask = mask ? static_cast<const uint16_t *>(mask[g]) : nullptr;
        a.g[g].colsum = colsum ? static_cast<float *>(colsum_ws) + (size_t)g * conv_igemm_pixel_tiles(B, H, W) * cout : nullptr;
        a.g[g].res = static_cast<const uint16_t *>(t_res);
        if (!a.g[g].x || !a.g[g].w || !a.g[g].y || (mask && !mask[g]) || (colsum && !colsum[g]))
            return set_error(DSRG_ERR_INVALID, "conv_igemm: null pointer");
    }
    a.out_scale = out_scale;
    a.xrow = cin * 2;
    if (t_split_cin > 0) {                                   // launch_conv_igemm_split: cin here is the VIRTUAL channel count 6 * real
        a.xrow = 3 * t_split_cin * 2;
        a.cpp = t_split_cin / 64;
        a.out_f32 = 1;
    }
    a.skip_taps = igemm_variant() != 6;                      // 6: tests / tools — every tap of every tile, as before round 5
    const bool fused_bwd = mask || colsum || t_res;
    if (t_res && (colsum || ngroups != 1 || drop_p != 0.0f || t_split_cin))
        return set_error(DSRG_ERR_UNSUPPORTED, "conv_igemm: a residual goes with one group, no column sums, no dropout");
    a.ngroups = ngroups; a.B = B; a.H = H; a.W = W; a.Cin = cin; a.Cout = cout; a.taps = k * k; a.relu = relu; a.M = (int)M;
    a.tiles_m = (int)((M + kBM - 1) / kBM);
    a.tiles_n = (cout + kBN - 1) / kBN;
    a.tiles_per_group = a.tiles_m * a.tiles_n;
    if (drop_p < 0.0f || drop_p >= 1.0f) return set_error(DSRG_ERR_INVALID, "conv_igemm: 0 <= dropout probability < 1");
    a.drop_thresh = (uint32_t)(drop_p * 256.0f + 0.5f);       // p is realised in steps of 1 / 256 (0.5 exactly)
    if (a.drop_thresh > 255) a.drop_thresh = 255;
    a.drop_scale = 256.0f / (float)(256 - (int)a.drop_thresh);
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
    static LdsGrant grant[3];
    const int variant = igemm_variant() == 2 ? 1 : 0;       // 1, 3: two stages of 64; 2: ring of four stages of 32
    a.stagger = igemm_variant() >= 3;
    const bool default_form = igemm_variant() == 3 || igemm_variant() == 8 || igemm_variant() == 9;      // (8 / 9: 3 with one tiling forced)
    const dim3 block(512);
    // stream-K only where it was measured to win (profiles/r04_igemm_stream_k.txt): a single round that fills at most 60 % of
    // the chip (conv4_1's data gradient: 106 tiles, 115 -> 88 us).  A cut tile costs its workgroups ~25 us (256 KB of
    // accumulators written through, read back, one acquire), which eats the sixth of the chip that 212 tiles leave idle (131 ->
    // 136 us), and launches of several rounds lose less to their last round than the round count suggests (workgroups of
    // different rounds overlap: fc6 x 4, 6.6 rounds, 810 us whole against 902 us dealt out).
    const int units = igemm_cus(), tiles_total = a.tiles_per_group * ngroups, nsteps = (cin / 64) * k * k;
    const bool sk_wins = tiles_total * 100 <= units * 60, sk_forced = igemm_variant() == 4;      // 4: tests / tools, wherever legal
    if (!fused_bwd && t_split_cin == 0 && cout % 128 == 0 && workspace && workspace_bytes >= conv_igemm_workspace() && ((default_form && sk_wins) || sk_forced) &&
        (long long)tiles_total * nsteps >= (long long)units * 8 && tiles_total * 3 >= units) {
        IgemmSkArgs sk;
        sk.base = a;
        sk.base.stagger = 1;
        sk.ws = static_cast<float *>(workspace);
        sk.flags = reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(workspace) + (size_t)units * (32 * 512 * 16));
        sk.units = units;
        sk.tiles_total = tiles_total;
        constexpr size_t lds = ICfg<64, 2>::LDS;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_igemm_sk_kernel), lds, grant[2])) return rc;
        DSRG_HIP_CHECK(hipMemsetAsync(sk.flags, 0, sizeof(uint32_t) * ((size_t)units + 1), stream));      // every launch (a graph replays it)
        hipLaunchKernelGGL(conv_igemm_sk_kernel, dim3(units), block, lds, stream, sk);
        DSRG_LAUNCH_CHECK();
        return DSRG_OK;
    }
    // tap skipping makes tiles unequal: spread every group's tiles over all XCDs (conv_igemm_body).  Only then: the layers
    // whose tiles all run the same steps keep the map they were tuned with (neighbouring pixel tiles share rows in one L2)
    for (int g = 0; g < ngroups; g++)
        if (a.skip_taps && k == 3 && a.g[g].dil >= 3) a.xcd_mix = 1;
    static const bool row_tiles_on = [] { const char *e = getenv("DSRG_IGEMM_ROW_TILES"); return !e || atoi(e) != 0; }();      // tools: A/B
    static const bool cls_tiles_on = [] { const char *e = getenv("DSRG_IGEMM_CLASS_TILES"); return !e || atoi(e) != 0; }();    // tools: A/B
    const bool rows_ok = a.xcd_mix && row_tiles_on && W <= kBM && conv_igemm_row_tiles(H, W);
    if (a.xcd_mix && cls_tiles_on && igemm_variant() != 8 && H <= 255 && W <= 255) {       // 8: tests — round 5's row-aligned tiles
        // the class order pays where its tiles run fewer K-steps than the tiling it replaces (a map of few tiles has most of them
        // straddle classes); decided once per geometry
        struct Key { int B, H, W, d[4], n, rows; bool operator<(const Key &o) const { return memcmp(this, &o, sizeof(Key)) < 0; } };
        static std::map<Key, bool> memo;
        static std::mutex memo_mutex;
        Key key;
        memset(&key, 0, sizeof(key));
        key.B = B; key.H = H; key.W = W; key.n = ngroups; key.rows = rows_ok;
        for (int g = 0; g < ngroups; g++) key.d[g] = a.g[g].dil;
        bool pays;
        {
            std::lock_guard<std::mutex> lock(memo_mutex);
            auto it = memo.find(key);
            if (it == memo.end()) {
                long long now = 0, then = 0;
                for (int g = 0; g < ngroups; g++) { now += tile_taps(B, H, W, a.g[g].dil, 2); then += tile_taps(B, H, W, a.g[g].dil, rows_ok ? 1 : 0); }
                it = memo.emplace(key, now * 100 < then * 97).first;
            }
            pays = it->second;
        }
        if (pays || igemm_variant() == 9) {                  // 9: tests — the class order wherever it is legal
            a.cls_tiles = 1;
            for (int g = 0; g < ngroups; g++) {
                HostClass c[kMaxClasses];
                const int n = build_classes(H, W, a.g[g].dil, c);
                long long q0 = 0;
                a.g[g].ncls = n;
                for (int k = 0; k < n; k++) {
                    a.g[g].cls[k].rect = (uint32_t)c[k].y0 | (uint32_t)(c[k].y1 - c[k].y0) << 8 | (uint32_t)c[k].x0 << 16 | (uint32_t)(c[k].x1 - c[k].x0) << 24;
                    a.g[g].cls[k].q0 = (int)q0;
                    q0 += (long long)B * (c[k].y1 - c[k].y0) * (c[k].x1 - c[k].x0);
                }
            }
        }
    }
    if (!a.cls_tiles && rows_ok) {
        // ... and cut them along map rows (see the kernel); tiles_m grows by the rows a band leaves empty (112 against 106 tiles
        // for sixteen 41x41 maps), which the skipped steps more than pay for
        a.row_tiles = 1;
        a.rows_per_tile = kBM / W;
        a.bands = (H + a.rows_per_tile - 1) / a.rows_per_tile;
        a.tiles_m = B * a.bands;
        a.tiles_per_group = a.tiles_m * a.tiles_n;
    }
    const dim3 grid(a.xcd_mix ? 8 * ngroups * ((a.tiles_m + 7) / 8) * a.tiles_n : a.tiles_per_group * ngroups);
    if (t_prep_d) {                                          // launch_conv_igemm_backward: arguments only, it launches the merged kernel
        *t_prep_d = a;
        *t_prep_grid = (int)grid.x;
        return DSRG_OK;
    }
    if (igemm_variant() == 5) {                              // early barrier (see conv_igemm_body)
        static LdsGrant grant_e;
        constexpr size_t lds = ICfg<64, 2>::LDS;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_igemm_kernel_64x2e), lds, grant_e)) return rc;
        hipLaunchKernelGGL(conv_igemm_kernel_64x2e, grid, block, lds, stream, a);
    } else if (variant == 0) {
        constexpr size_t lds = ICfg<64, 2>::LDS;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_igemm_kernel_64x2), lds, grant[0])) return rc;
        hipLaunchKernelGGL(conv_igemm_kernel_64x2, grid, block, lds, stream, a);
    } else {
        constexpr size_t lds = ICfg<32, 4>::LDS;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_igemm_kernel_32x4), lds, grant[1])) return rc;
        hipLaunchKernelGGL(conv_igemm_kernel_32x4, grid, block, lds, stream, a);
    }
    DSRG_LAUNCH_CHECK();
    if (colsum) {
        const float *parts[4];
        for (int g = 0; g < ngroups; g++) parts[g] = a.g[g].colsum;
        if (int rc = launch_igemm_colsum(parts, colsum, ngroups, a.tiles_m, cout, stream)) return rc;
    }
    return DSRG_OK;
}


// A float32 convolution on the bf16 MFMA (forward, one group; layer-level prototype of round 6): x3 = the three bf16 planes of the
// float32 activation, (B, H, W, 3 cin) with channel index plane * cin + c; w = the packed kernel over 6 cin VIRTUAL channels,
// (cout, 6 cin / 64, k k, 64), virtual chunk block q holding plane {0, 1, 0, 2, 0, 1}[q] of the split kernel; y float32 (B, H, W, cout)
int launch_conv_igemm_split(const void *x3, const void *w, const float *bias, float *y, int dil, int B, int H, int W, int cin, int cout, int k,
                            int relu, hipStream_t stream) {
    if (!conv_igemm_supported(cin, cout, k) || (long long)B * H * W * 3 * cin * 2 >= 0x7fffffffLL)
        return set_error(DSRG_ERR_UNSUPPORTED, "conv_igemm_split: shape not supported");
    const void *xp[1] = {x3}, *wp[1] = {w};
    const float *bp[1] = {bias};
    void *yp[1] = {y};
    const int dils[1] = {dil};
    t_split_cin = cin;
    const int rc = launch_conv_igemm(xp, wp, bias ? bp : nullptr, yp, dils, 1, B, H, W, 6 * cin, cout, k, relu, 0.0f, 0ull, nullptr, 0, stream,
                                     nullptr, 1.0f, nullptr, nullptr, 0);
    t_split_cin = 0;
    return rc;
}

// One group with a residual in the store (IgemmGroup::res): y = post(bf16(conv(x, w) + bias) + res), post = ReLU (relu) and / or
// zero where mask <= 0 (mask may be nullptr).  Forward of a residual block's last convolution; data gradient of its first.
int launch_conv_igemm_residual(const void *x, const void *w, const float *bias, const void *res, const void *mask, void *y, int dil, int B,
                               int H, int W, int cin, int cout, int k, int relu, hipStream_t stream) {
    if (!res) return set_error(DSRG_ERR_INVALID, "conv_igemm_residual: null residual");
    const void *xp[1] = {x}, *wp[1] = {w}, *mp[1] = {mask};
    const float *bp[1] = {bias};
    void *yp[1] = {y};
    const int dils[1] = {dil};
    t_res = res;
    const int rc = launch_conv_igemm(xp, wp, bias ? bp : nullptr, yp, dils, 1, B, H, W, cin, cout, k, relu, 0.0f, 0ull, nullptr, 0, stream,
                                     mask ? mp : nullptr, 1.0f, nullptr, nullptr, 0);
    t_res = nullptr;
    return rc;
}

bool conv_igemm_wgrad_supported(int cin, int cout, int k) {
    return (k == 1 || k == 3) && ((cin >= 256 && cin % 256 == 0) || (cin == 128 && k == 3)) && cout >= 256 && cout % 256 == 0;
}
// what the launch takes (conv_igemm_wgrad_supported: where it is the recommended route): any multiples of 64 channels — a tile is 256 outputs
// x (one tap x 256 inputs), narrower tensors leave part of it empty (ResNet res2 / res3: 64 / 128 channels over 42 - 166 thousand pixels,
// bandwidth-bound either way)
static bool conv_igemm_wgrad_launchable(int cin, int cout, int k) {
    return (k == 1 || k == 3) && cin >= 64 && cin % 64 == 0 && cout >= 64 && cout % 64 == 0;
}
static int wgrad_col_tiles(int cin, int k) { return (cin == 128 && k == 3) ? (k * k + 1) / 2 : k * k * ((cin + 255) / 256); }

// pixel split of the weight-gradient launch: the number of workgroups per output tile that minimises
// rounds of the chip x (K-steps per workgroup + a fixed cost per workgroup for prologue and the partial tile's write-out)
// out_bytes: the gradient tensors of all groups — every split writes and the reduction reads that much again, ~2.3 us (one
// K-step of a workgroup) per 9.2 MB at the rate the reduction kernel streams (the four fc6_k: 75 MB, 8 K-steps per split)
// xcd_mix_only: only the splits the XCD-interleaved workgroup map takes (a divisor or a multiple of 8)
static int wgrad_ksplit(long long M, int tiles, int cus, double out_bytes = 0.0, bool xcd_mix_only = false) {
    long long best_cost = -1;
    int best = 1;
    static const int forced = [] { const char *e = getenv("DSRG_WGRAD_KSPLIT"); return e ? atoi(e) : 0; }();     // tools only
    if (forced >= 1 && forced <= 128 && (long long)(forced - 1) * (((M + forced - 1) / forced + 63) / 64 * 64) < M) return forced;
    for (int ks = 1; ks <= 128; ks++) {
        const long long chunk = ((M + ks - 1) / ks + 63) / 64 * 64;
        if ((long long)(ks - 1) * chunk >= M) continue;                 // an empty last split
        if (xcd_mix_only && ks % 8 != 0 && 8 % ks != 0) continue;
        const long long steps = chunk / 64, rounds = ((long long)tiles * ks + cus - 1) / cus;
        const long long cost = rounds * (steps + 8) + (long long)(ks * out_bytes / 9.2e6);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = ks; }
    }
    return best;
}

// the finest pixel split launch_conv_igemm_backward may pick for a layer whose stand-alone split is ks: twice as fine, never beyond
// one 64-pixel step per split
static int wgrad_ksplit_cap(long long M, int ks) {
    const long long most = (M + 63) / 64;
    long long c = 2LL * ks;
    if (c > most) c = most;
    if (c > 128) c = 128;
    return (int)(c < ks ? ks : c);
}

size_t conv_igemm_wgrad_workspace(int ngroups, int B, int H, int W, int cin, int cout, int k) {
    if (!conv_igemm_wgrad_launchable(cin, cout, k) || ngroups < 1 || ngroups > 4) return 0;
    const long long M = (long long)B * H * W;
    const int tiles = ngroups * ((cout + 255) / 256) * wgrad_col_tiles(cin, k);
    int ks = wgrad_ksplit(M, tiles, 256, (double)ngroups * cout * k * k * cin * 4.0);
    const int ks_mix = wgrad_ksplit(M, tiles, 256, (double)ngroups * cout * k * k * cin * 4.0, true);      // (a launch of dilated kernels picks among these)
    if (ks_mix > ks) ks = ks_mix;
    if (ngroups == 1) ks = wgrad_ksplit_cap(M, ks);         // (the merged backward launch may cut a single layer's pixels finer)
    return (size_t)ngroups * ks * cout * k * k * cin * sizeof(float);
}

static int launch_wgrad_reduce(const IgemmWgradArgs &a, void *const *gw, int ngroups, int cin, int cout, int k, int out_bf16, hipStream_t stream) {
    WgradReduceArgs r;
    memset(&r, 0, sizeof(r));
    r.ksplit = a.ksplit;
    r.n4 = (size_t)cout * k * k * cin / 4;
    r.scale = t_wgrad_scale;
    r.per_o4 = (size_t)k * k * cin / 4;
    for (int q = 0; q < ngroups; q++) { r.part[q] = a.g[q].part; r.gw[q] = gw[q]; }
    const dim3 rgrid((unsigned)((r.n4 + 255) / 256), (unsigned)ngroups);
    if (out_bf16) hipLaunchKernelGGL(conv_igemm_wgrad_reduce_kernel<true>, rgrid, dim3(256), 0, stream, r);
    else hipLaunchKernelGGL(conv_igemm_wgrad_reduce_kernel<false>, rgrid, dim3(256), 0, stream, r);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

int launch_conv_igemm_wgrad(const void *const *x, const void *const *g, void *const *gw, const int *dil, int ngroups, void *workspace,
                            size_t workspace_bytes, int B, int H, int W, int cin, int cout, int k, int out_bf16, hipStream_t stream) {
    if (ngroups < 1 || ngroups > 4) return set_error(DSRG_ERR_INVALID, "conv_igemm_wgrad: 1..4 groups");
    if (!conv_igemm_wgrad_launchable(cin, cout, k))
        return set_error(DSRG_ERR_UNSUPPORTED, "conv_igemm_wgrad: cin %% 64 == 0, cout %% 64 == 0, k in (1, 3) required (got %d, %d, %d)", cin, cout, k);
    const long long M = (long long)B * H * W;
    if (M <= 0 || M * cin * 2 >= 0x7fffffffLL || M * cout * 2 >= 0x7fffffffLL)
        return set_error(DSRG_ERR_UNSUPPORTED, "conv_igemm_wgrad: tensor too large for 32-bit buffer offsets");
    const size_t need = conv_igemm_wgrad_workspace(ngroups, B, H, W, cin, cout, k);
    if (!workspace || workspace_bytes < need) return set_error(DSRG_ERR_INVALID, "conv_igemm_wgrad: workspace of %zu bytes needed", need);
    IgemmWgradArgs a;
    memset(&a, 0, sizeof(a));
    a.ngroups = ngroups; a.B = B; a.H = H; a.W = W; a.Cin = cin; a.Cout = cout; a.taps = k * k; a.M = (int)M;
    a.tiles_n = (cout + 255) / 256;
    a.tiles_c = wgrad_col_tiles(cin, k);
    bool wants_mix = false;                                  // dilated kernels: workgroups of unequal length, see xcd_mix
    for (int q = 0; q < ngroups; q++) wants_mix = wants_mix || (k == 3 && dil && dil[q] >= 3);      // (whatever the variant: tests compare them bit for bit)
    a.ksplit = t_force_ksplit > 0 ? t_force_ksplit
                                  : wgrad_ksplit(M, ngroups * a.tiles_n * a.tiles_c, 256, (double)ngroups * cout * k * k * cin * 4.0, wants_mix);
    static const int grouped_ks = [] { const char *e = getenv("DSRG_WGRAD_KSPLIT_GROUPED"); return e ? atoi(e) : 0; }();        // tools: A/B
    if (grouped_ks > 0 && ngroups == 4 && k == 3 && grouped_ks <= a.ksplit) a.ksplit = grouped_ks;
    a.kchunk = (int)(((M + a.ksplit - 1) / a.ksplit + 63) / 64 * 64);
    a.tiles_per_group = a.tiles_n * a.tiles_c * a.ksplit;
    a.stagger = igemm_variant() >= 3;
    a.skip_rows = igemm_variant() != 6;
    const size_t per_group = (size_t)a.ksplit * cout * k * k * cin;
    for (int q = 0; q < ngroups; q++) {
        a.g[q].x = static_cast<const uint16_t *>(x[q]);
        a.g[q].g = static_cast<const uint16_t *>(g[q]);
        a.g[q].part = static_cast<float *>(workspace) + (size_t)q * per_group;
        a.g[q].dil = dil ? dil[q] : 1;
        if (!a.g[q].x || !a.g[q].g || !gw[q]) return set_error(DSRG_ERR_INVALID, "conv_igemm_wgrad: null pointer");
    }
    static const bool compact_on = [] { const char *e = getenv("DSRG_WGRAD_COMPACT"); return !e || atoi(e) != 0; }();      // tools: A/B
    static const int compact_min_dil = [] { const char *e = getenv("DSRG_WGRAD_COMPACT_MIN_DIL"); return e ? atoi(e) : 1; }();     // tools: A/B (3: dilated kernels only — 1 776-1 782 against 1 800-1 807 images/s)
    for (int q = 0; q < ngroups; q++) {
        if (a.skip_rows && k == 3 && a.g[q].dil >= 3 &&
            (a.ksplit % 8 == 0 || (8 % a.ksplit == 0 && (a.tiles_n * a.tiles_c) % (8 / a.ksplit) == 0))) a.xcd_mix = 1;
        if (a.skip_rows && compact_on && igemm_variant() != 7 && k == 3 && cin != 128 && a.g[q].dil >= compact_min_dil) a.compact = 1;      // 7: tests — dead steps skipped in the flat pixel order
    }
    if (t_prep_w) {
        *t_prep_w = a;
        *t_prep_grid = a.tiles_per_group * ngroups;
    } else {
        static LdsGrant grant;
        constexpr size_t lds = 2 * kWStage;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_igemm_wgrad_kernel), lds, grant)) return rc;
        hipLaunchKernelGGL(conv_igemm_wgrad_kernel, dim3(a.tiles_per_group * ngroups), dim3(512), lds, stream, a);
        DSRG_LAUNCH_CHECK();
    }
    if (t_prep_w) return DSRG_OK;                            // (the caller runs the reduction below itself, after its merged launch)
    return launch_wgrad_reduce(a, gw, ngroups, cin, cout, k, out_bf16, stream);
}


// when the last block of a merged backward grid ends (us): blocks go to the 8 XCDs round-robin by id, an XCD's CUs take its
// blocks in id order as they free up — nd data-gradient tiles of td us first, then (from the next multiple of 8) nw weight-gradient
// workgroups of tw us
static double merged_makespan(int nd, double td, int nw, double tw) {
    const int per_xcd = igemm_cus() / 8 > 0 ? igemm_cus() / 8 : 32;
    double worst = 0.0;
    for (int x = 0; x < 8; x++) {
        std::priority_queue<double, std::vector<double>, std::greater<double>> cu;
        for (int c = 0; c < per_xcd; c++) cu.push(0.0);
        const int ndx = (nd - x + 7) / 8 > 0 ? (nd - x + 7) / 8 : 0, nwx = (nw - x + 7) / 8 > 0 ? (nw - x + 7) / 8 : 0;
        double end = 0.0;
        for (int b = 0; b < ndx + nwx; b++) {
            const double t = cu.top() + (b < ndx ? td : tw);
            cu.pop();
            cu.push(t);
            if (t > end) end = t;
        }
        if (end > worst) worst = end;
    }
    return worst;
}

// The backward of ONE 3x3 convolution as one launch (conv_igemm_bwd_kernel): data gradient of g with the flipped kernel `wd` (+ the
// ReLU / Dropout backward and bias gradient of the layer below when `mask` is given, as launch_conv_igemm's fused form) and the
// weight gradient from (x, g).  Falls back to the two launches where the merged kernel does not apply (grouped launches, the
// XCD-interleaved or stream-K forms, a 128-channel x): same results either way — the two halves run the very code of the
// separate kernels on the very argument blocks.
int launch_conv_igemm_backward(const void *g, const void *wd, const void *x, const void *mask, void *gx, void *gw, int dil, float *bias_grad,
                               float mask_scale, void *colsum_ws, size_t colsum_ws_bytes, void *wgrad_ws, size_t wgrad_ws_bytes, int B, int H,
                               int W, int cin, int cout, int k, hipStream_t stream) {
    // (forward convolution: cin -> cout; g has cout channels, gx and x have cin, gw is (cout, cin, k, k) in float32)
    const void *gp[1] = {g}, *wp[1] = {wd}, *xp[1] = {x}, *mp[1] = {mask};
    void *gxp[1] = {gx}, *gwp[1] = {gw};
    float *bgp[1] = {bias_grad};
    const int dils[1] = {dil};
    static const bool merged_on = [] { const char *e = getenv("DSRG_IGEMM_MERGED_BWD"); return !e || atoi(e) != 0; }();      // tools: A/B
    static const bool w_first_on = [] { const char *e = getenv("DSRG_MERGED_W_FIRST"); return !e || atoi(e) != 0; }();      // tools: A/B
    static const bool merged_k1_on = [] { const char *e = getenv("DSRG_IGEMM_MERGED_K1"); return !e || atoi(e) != 0; }();    // tools: A/B (1x1 layers)
    const bool can_merge = merged_on && ((k == 3 && dil < 3) || (k == 1 && merged_k1_on)) && (igemm_variant() == 3 || igemm_variant() == 1 || igemm_variant() == 8 || igemm_variant() == 9);
    IgemmBwdArgs a;
    memset(&a, 0, sizeof(a));
    int nd = 0, nw = 0, rc = DSRG_OK;
    if (can_merge) {
        t_prep_d = &a.d; t_prep_grid = &nd;
        rc = launch_conv_igemm(gp, wp, nullptr, gxp, dils, 1, B, H, W, cout, cin, k, 0, 0.0f, 0ull, nullptr, 0, stream, mask ? mp : nullptr,
                               mask_scale, bias_grad ? bgp : nullptr, colsum_ws, colsum_ws_bytes);
        t_prep_d = nullptr;
        if (!rc && !a.d.xcd_mix && nd > 0) {
            // the pixel split of the weight gradient half, chosen for THIS grid: its workgroups fill the CUs the data gradient's
            // tiles leave idle and then the whole chip; what counts is when the last of them ends (merged_makespan)
            const long long M = (long long)B * H * W;
            const int tiles = ((cout + 255) / 256) * wgrad_col_tiles(cin, k), ks0 = wgrad_ksplit(M, tiles, 256, (double)cout * k * k * cin * 4.0), cap = wgrad_ksplit_cap(M, ks0);
            const double td = ((cout / 64) * k * k + 6) * 1.85;                  // us per data-gradient tile: K-steps + prologue / epilogue
            double best = -1.0;
            int best_ks = ks0;
            // the search simulates up to 2 x 128 grids of ~1 000 blocks — 0.25 - 0.5 ms of host time, as much as the launch runs on the
            // GPU: decided once per geometry (a ResNet-101 step has 77 of these launches)
            struct Key { int B, H, W, cin, cout, k, nd; bool operator<(const Key &o) const { return memcmp(this, &o, sizeof(Key)) < 0; } };
            static std::map<Key, std::pair<int, int>> memo;
            static std::mutex memo_mutex;
            Key key;
            memset(&key, 0, sizeof(key));
            key.B = B; key.H = H; key.W = W; key.cin = cin; key.cout = cout; key.k = k; key.nd = nd;
            bool known = false;
            {
                std::lock_guard<std::mutex> lock(memo_mutex);
                auto it = memo.find(key);
                if (it != memo.end()) { best_ks = it->second.first; a.w_first = it->second.second; known = true; }
            }
            // (the block order is searched too for the shapes round 6 added — 1x1 layers, channel counts below 256: their data-gradient
            // tiles are a few K-steps long, and long weight-gradient workgroups dispatched LAST would run on alone; the 3x3 layers of
            // the VGG path keep the order they were tuned with)
            static const bool order_all = [] { const char *e = getenv("DSRG_MERGED_ORDER_ALL"); return e && atoi(e) != 0; }();     // tools: A/B
            const bool order_free = w_first_on && (order_all || k == 1 || cin < 256 || cout < 256);
            for (int ks = 1; ks <= cap && !known; ks++) {
                const long long chunk = ((M + ks - 1) / ks + 63) / 64 * 64;
                if ((long long)(ks - 1) * chunk >= M) continue;                  // an empty last split
                const double tw = (chunk / 64 + 8) * 2.3;                        // us per weight-gradient workgroup: steps + partial tile out
                const double tail = 3.0 + 2.4 * ks * ((double)cout * k * k * cin / (512.0 * 4608.0));
                const double t = merged_makespan(nd, td, tiles * ks, tw) + tail;
                if (best < 0.0 || t < best) { best = t; best_ks = ks; a.w_first = 0; }
                if (order_free) {
                    const double t2 = merged_makespan(tiles * ks, tw, nd, td) + tail;
                    if (t2 < best) { best = t2; best_ks = ks; a.w_first = 1; }
                }
            }
            if (!known) {
                std::lock_guard<std::mutex> lock(memo_mutex);
                memo[key] = std::make_pair(best_ks, a.w_first);
            }
            static const bool pick_on = [] { const char *e = getenv("DSRG_MERGED_KS"); return !e || atoi(e) != 0; }();      // tools: A/B
            t_force_ksplit = pick_on ? best_ks : 0;
            t_prep_w = &a.w; t_prep_grid = &nw;
            rc = launch_conv_igemm_wgrad(xp, gp, gwp, dils, 1, wgrad_ws, wgrad_ws_bytes, B, H, W, cin, cout, k, 0, stream);
            t_prep_w = nullptr;
            t_force_ksplit = 0;
        }
        t_prep_grid = nullptr;
        if (rc) return rc;
    }
    if (!can_merge || a.d.xcd_mix || nd < 1 || nw < 1) {
        rc = launch_conv_igemm(gp, wp, nullptr, gxp, dils, 1, B, H, W, cout, cin, k, 0, 0.0f, 0ull, nullptr, 0, stream, mask ? mp : nullptr,
                               mask_scale, bias_grad ? bgp : nullptr, colsum_ws, colsum_ws_bytes);
        if (rc) return rc;
        return launch_conv_igemm_wgrad(xp, gp, gwp, dils, 1, wgrad_ws, wgrad_ws_bytes, B, H, W, cin, cout, k, 0, stream);
    }
    a.nd = nd; a.nd_pad = (nd + 7) & ~7; a.nw = nw; a.nw_pad = (nw + 7) & ~7;
    static LdsGrant grant;
    constexpr size_t lds = ICfg<64, 2>::LDS > (size_t)(2 * kWStage) ? ICfg<64, 2>::LDS : (size_t)(2 * kWStage);
    if (int rc2 = ensure_dynamic_lds(reinterpret_cast<const void *>(&conv_igemm_bwd_kernel), lds, grant)) return rc2;
    hipLaunchKernelGGL(conv_igemm_bwd_kernel, dim3(a.w_first ? a.nw_pad + a.nd : a.nd_pad + a.nw), dim3(512), lds, stream, a);
    DSRG_LAUNCH_CHECK();
    if (bias_grad) {
        const float *parts[1] = {a.d.g[0].colsum};
        if (int rc2 = launch_igemm_colsum(parts, bgp, 1, a.d.tiles_m, cin, stream)) return rc2;
    }
    return launch_wgrad_reduce(a.w, gwp, 1, cin, cout, k, 0, stream);
}


int launch_pack_conv_weight(const float *w, void *fwd, void *dgrad, int cout, int cin, int k, hipStream_t stream, int plain, const float *scale) {
    if (!w || cout < 64 || cout % 64 || cin < 64 || cin % 64 || (k != 1 && k != 3))
        return set_error(DSRG_ERR_INVALID, "pack_conv_weight: 64 | cout, 64 | cin, k in (1, 3) required (got %d, %d, %d)", cout, cin, k);
    if (!fwd && !dgrad) return DSRG_OK;
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(cout / 64, cin / 64, k * k), dim3(256), 0, stream, w, static_cast<uint16_t *>(fwd),
                       static_cast<uint16_t *>(dgrad), cout, cin, k * k, plain, scale);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

// launch_conv_igemm_backward with a residual in the data gradient's store (IgemmGroup::res; res may be nullptr) and a per-output-channel
// factor on the weight gradient (gw_scale may be nullptr); no bias gradient.  A ResNet bottleneck convolution's whole backward.
int launch_conv_igemm_backward_residual(const void *g, const void *wd, const void *x, const void *mask, const void *res, void *gx, void *gw,
                                        const float *gw_scale, int dil, void *wgrad_ws, size_t wgrad_ws_bytes, int B, int H, int W, int cin,
                                        int cout, int k, hipStream_t stream) {
    t_res = res;
    t_wgrad_scale = gw_scale;
    const int rc = launch_conv_igemm_backward(g, wd, x, mask, gx, gw, dil, nullptr, 1.0f, nullptr, 0, wgrad_ws, wgrad_ws_bytes, B, H, W, cin, cout,
                                              k, stream);
    t_res = nullptr;
    t_wgrad_scale = nullptr;
    return rc;
}

// tests: the error word of the last stream-K launch that used this workspace (1 = a workgroup gave up waiting); synchronises
int conv_igemm_workspace_status(const void *workspace, hipStream_t stream, int *status) {
    uint32_t v = 0;
    const unsigned char *p = static_cast<const unsigned char *>(workspace) + (size_t)igemm_cus() * (32 * 512 * 16) + sizeof(uint32_t) * (size_t)igemm_cus();
    DSRG_HIP_CHECK(hipMemcpyAsync(&v, p, sizeof(v), hipMemcpyDeviceToHost, stream));
    DSRG_HIP_CHECK(hipStreamSynchronize(stream));
    *status = (int)v;
    return DSRG_OK;
}

}  // namespace dsrg
