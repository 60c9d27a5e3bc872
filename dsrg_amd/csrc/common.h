// Internal declarations shared by the HIP translation units of libdsrg_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>
#include "../../include/dsrg_hip.h"

namespace dsrg {

// ---- error plumbing --------------------------------------------------------
int set_error(int code, const char *fmt, ...);
#define DSRG_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return ::dsrg::set_error(DSRG_ERR_HIP, "%s failed: %s (%s:%d)", #expr,        \
                                     hipGetErrorString(_e), __FILE__, __LINE__);          \
    } while (0)
#define DSRG_LAUNCH_CHECK() DSRG_HIP_CHECK(hipGetLastError())

// ---- the small reductions that finish a bias gradient (column sums of a launch's per-tile partial rows; the direct kernels' per-block
// partial rows), as device bodies shared by their own kernels and by the deferred multi-reduction (deferred.hip): same arithmetic,
// same order, whoever runs them.
// column sums of `rows` partial rows of `cout` floats: block bx owns channels [64 bx, 64 bx + 64), 1024 threads = 16 row slices
__device__ __forceinline__ void colsum_block(const float *__restrict__ p, float *__restrict__ out, int rows, int cout, int bx, float (*red)[64]) {
    const int cl = threadIdx.x & 63, c = bx * 64 + cl, q = threadIdx.x >> 6;
    float s0 = 0.0f, s1 = 0.0f;
    int r = q;
    for (; r + 16 < rows; r += 32) {
        s0 += p[(size_t)r * cout + c];
        s1 += p[(size_t)(r + 16) * cout + c];
    }
    if (r < rows) s0 += p[(size_t)r * cout + c];
    red[q][cl] = s0 + s1;
    __syncthreads();
    if (q == 0) {
        float t[4];
#pragma unroll
        for (int k = 0; k < 4; k++) t[k] = (red[4 * k][cl] + red[4 * k + 1][cl]) + (red[4 * k + 2][cl] + red[4 * k + 3][cl]);
        out[c] = (t[0] + t[1]) + (t[2] + t[3]);
    }
}
// 32 channels x 8 slices of `nblk` partial rows of C floats per block (threads 0-255; further threads of a larger block idle along);
// slice sums are combined in slice order
__device__ __forceinline__ void bias_finalize_block(const float *__restrict__ part, float *__restrict__ bias_grad, int nblk, int C, int bx,
                                                    float (*red)[33]) {
    const bool on = threadIdx.x < 256;
    const int cl = threadIdx.x & 31, sl = (threadIdx.x >> 5) & 7;
    const int c = bx * 32 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (on && c < C) {
        int b = sl;
        for (; b + 24 < nblk; b += 32) {
            s0 += part[(size_t)b * C + c];
            s1 += part[(size_t)(b + 8) * C + c];
            s2 += part[(size_t)(b + 16) * C + c];
            s3 += part[(size_t)(b + 24) * C + c];
        }
        for (; b < nblk; b += 8) s0 += part[(size_t)b * C + c];
    }
    if (on) red[sl][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (on && sl == 0 && c < C) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += red[k][cl];
        bias_grad[c] = s;
    }
}
// Deferred form (dsrg_defer_reductions / dsrg_flush_reductions): while deferral is on, a launch that would end with one of the two
// passes above records it instead — true: recorded, the caller launches nothing; false: not deferring (or the list is full).
// kind 0: colsum_block over (rows, cols = cout); kind 1: bias_finalize_block over (rows = nblk, cols = C).
bool defer_reduction(int kind, const float *part, float *out, int rows, int cols);

// raise a kernel's dynamic-LDS limit (default 64 KiB) to `bytes`; `granted` caches what was set.
// The limit is requested per need, not as a flat 160 KiB: static LDS (e.g. the variable behind
// __syncthreads_or) counts against the same 160 KiB and an over-ask is rejected.
// The cache is shared by every host thread that launches the kernel (function-static tables): entries are atomics, and a
// race between two threads costs at most a repeated hipFuncSetAttribute with the larger of their sizes winning last —
// hipFuncSetAttribute itself is thread-safe, and a launch never asks for more than its own thread has just reserved
// because the value only ever grows (compare-and-swap to the maximum).
struct LdsGrant { std::atomic<size_t> bytes[16]; LdsGrant() { for (auto &b : bytes) b.store(0, std::memory_order_relaxed); } };
inline int ensure_dynamic_lds(const void *fn, size_t bytes, std::atomic<size_t> &granted) {
    size_t have = granted.load(std::memory_order_acquire);
    if (bytes <= have) return DSRG_OK;
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess)
            return set_error(DSRG_ERR_HIP, "cannot reserve %zu B of dynamic LDS: %s", bytes, hipGetErrorString(e));
    }
    while (have < bytes && !granted.compare_exchange_weak(have, bytes, std::memory_order_release, std::memory_order_acquire)) {}
    return DSRG_OK;
}
inline int ensure_dynamic_lds(const void *fn, size_t bytes, LdsGrant &g) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { std::atomic<size_t> none(0); return ensure_dynamic_lds(fn, bytes, none); }
    return ensure_dynamic_lds(fn, bytes, g.bytes[dev]);
}

constexpr int kWG = 1024;          // threads per workgroup of the lattice kernels (16 waves)
constexpr int kMaxLabels = 96;     // per-pixel label loops keep at most this many values in registers / LDS columns (the
                                   // 81-class blobs of AnnotationLayerCOCO, pylayers.py:389-512, fit)
constexpr float kMinProb = 0.0001f;

// ---- permutohedral lattice, device-resident ----------------------------------
// One lattice = one (image, kernel) pair.  All arrays are sized for the worst
// case Mcap = Npad*(d+1) vertices so nothing depends on the data-dependent M.
// Vertex ids and pixel indices fit uint16 on this path (Mcap+1, N < 65536).
struct LatticeView {
    int d;                 // feature dimension (2 Gaussian, 5 bilateral)
    int N;                 // pixels
    int Mcap;              // capacity = ((N+3)/4*4)*(d+1); value slot M (<= Mcap) is the zero sentinel
    int nlat;              // number of lattices in this set (1 for the shared Gaussian, B for bilateral)
    // per lattice, strided by the quantities in brackets
    int *M;                // [1]        vertex count
    int *flags;            // [1]        bit 0: lattice is diagonal (every vertex has one contributor, no blur neighbour);
                           //            bit 2: every vertex has exactly one contributor (M == E);
                           //            bit 4 (kLatticeLocal, d = 2 only): pixel-local — see loc_a
    uint16_t *vid;         // [(d+1)*N]  vertex id of simplex corner r of pixel i   (r-major)
    float *bary;           // [(d+1)*N]  barycentric weight                         (r-major)
    uint32_t *nb;          // [(d+1)*Mcap] blur neighbours along axis j: n1 | n2<<16; M (the zero sentinel slot) = none, and
                           //               so is every word of the unused tail v >= M
    uint16_t *row_start;   // [Mcap+2]   CSR of the splat: entries of vertex v (E < 65536 on this path)
    float *csr_w;          // [(d+1)*N]  weight of each entry, entry order = reference splat order (the norm pass reads it)
    // the filter kernel's form of the splat: the first contributor of every vertex, vertex-indexed, and all further
    // entries ("extras") as one compact list; the extras of row v are [row_start[v] - v, row_start[v+1] - v - 1)
    uint16_t *first_pix;   // [Mcap]     source pixel of vertex v's first entry (0 for a vertex without entries)
    float *first_w;        // [Mcap]     its weight (0 for a vertex without entries)
    uint16_t *x_pix;       // [(d+1)*N]  source pixel of extra entry t
    float *x_w;            // [(d+1)*N]  its weight
    int *nextra;           // [1]        number of extras = E - (vertices with at least one entry)
    float *norm;           // [N]        1/sqrt(K 1 + 1e-20)
    // d = 2 only, valid when flag kLatticeLocal is set: every vertex has one contributor and the only blur neighbours of a
    // pixel's three corners are each other (the spatial kernel at training scale, sigma = 0.25 px).  Corners relabelled per
    // pixel so that blur axis j exchanges between relabelled corners (j, j+1 mod 3); see meanfield.hip, GaussLocal
    float *loc_a;          // [4*N]      (norm, weight of relabelled corner 0, 1, 2)
    uint32_t *loc_z;       // [N]        relabelled index of original corner 2 (the last term of the slice sum)
    // build scratch
    uint32_t *key_e;       // [Epad*KW]  packed keys of every (pixel, corner) entry, Epad = Npad*(d+1)
    uint16_t *slot_e;      // [Epad]     hash slot of every entry
    uint32_t *key_v;       // [Mcap*KW]  packed key of every vertex
    uint32_t *tab_g;       // [cap/2]    hash table (16-bit slots) handed from the build to the neighbour kernel
    unsigned long long *ckeys_g;   // [Mcap]   compact vertex keys, likewise
    unsigned long long *ckeys_e;   // [Epad]   compact keys of every entry, handed from the embedding kernel to the build
    int *embed_bad;        // [32]       per workgroup of the embedding kernel: bit 0 a coordinate beyond the short range,
                           //            bit 1 beyond the 12-bit compact range
};

// where the colours of the bilateral lattices' pixels come from: a (nlat, N, 3) uint8 image, or the mean-subtracted float
// images of the net resampled to the map (pylayers.py:70-75) on the fly — the uint8 image is then also written to im_out
struct LatticeColours {
    const unsigned char *im_u8 = nullptr;
    const float *images = nullptr;
    int Hi = 0, Wi = 0;
    unsigned char *im_out = nullptr;
};

struct LatticeFeat {
    // feature = coordinate / sigma  (densecrf.cpp:61-81); fp32 divisions
    float sx, sy, sr, sg, sb;
    float scale[5];        // E's diagonal (permutohedral.cpp:179-182), evaluated on the host
    int W, H;
};

size_t lattice_bytes(int d, int N, int nlat);
// carve a LatticeView out of one allocation (base must be 256-B aligned)
void lattice_carve(LatticeView &L, void *base, int d, int N, int nlat);
void lattice_feat_init(LatticeFeat &F, int d, int W, int H, float sx, float sy, float sr, float sg, float sb);

// builds `nlat` lattices (no colours for the Gaussian)
int launch_lattice_build(const LatticeView &L, const LatticeFeat &F, const LatticeColours &col, int nlat,
                         hipStream_t stream);
bool lattice_supported(int d, int N);

// ---- mean field -------------------------------------------------------------------
struct MeanfieldBufs {
    float *q;        // (B,C,N) current marginals
    float *msg_g;    // (B,C,N) normalised Gaussian message  K~_g Q (unused while the Gaussian lattice is pixel-local)
    float *msg_b;    // (B,C,N) normalised bilateral message K~_b Q
};
constexpr int kLatticeLocal = 16;      // LatticeView::flags bit 4
// optional per-launch timing of the filter kernel with HIP events on the launch stream
struct Profiler {
    hipEvent_t *start, *stop;   // [cap]
    int cap, used;
    bool active;
};
// gauss_local: the host has read the Gaussian lattice's flags and found kLatticeLocal (its workgroups are then left out of
// the filter launches; without that knowledge they are launched and exit on the device-side flag)
int launch_meanfield(const LatticeView &Lg, const LatticeView &Lb, const MeanfieldBufs &buf, int B, int C,
                     const float *neg_unary, float wg, float wb, int n_iters, float *q_out,
                     double *refined_out, float *logq_out, bool gauss_local, hipStream_t stream, Profiler *prof = nullptr,
                     bool q0_ready = false);     // q0_ready: buf.q already holds Q0 = expAndNormalize(neg_unary) (n_iters > 0)
int filter_plan_query(const LatticeView &Lg, const LatticeView &Lb, const MeanfieldBufs &buf, int B, int C, bool gauss_local,
                      int out[4]);
int launch_lattice_norm_pass(const LatticeView &L, int nlat, hipStream_t stream);      // meanfield.hip; d = 5 lattices
int launch_filter_once(const LatticeView &Lg, const LatticeView &Lb, const MeanfieldBufs &buf, int B, int C, int kind,
                       const float *q_in, float *out, bool gauss_local, hipStream_t stream);

// ---- pointwise / prep ----------------------------------------------------------------
int launch_clip_min(float *p, size_t n, hipStream_t stream);

// q0 (optional): also write expAndNormalize(p), the mean field's starting point when the unary energy is -p (CRF.py:28)
int launch_softmax_fwd(int B, int C, int HW, const float *x, float *p, hipStream_t stream, float floor_at = 0.0f,
                       float *q0 = nullptr);
int launch_softmax_bwd(int B, int C, int HW, const float *x, const float *g, float *dx, hipStream_t stream);
int launch_crf_bwd(size_t n, const double *refined, const float *td, float *bd, hipStream_t stream);
int launch_seed_loss(int B, int C, int HW, const float *p, const float *S, float *loss, float *grad,
                     hipStream_t stream);
int launch_constrain_loss(int B, int C, int HW, const float *p, const float *lq, float *loss, float *gp,
                          float *glq, hipStream_t stream);
int launch_sup_loss_backward(int B, int C, int HW, const float *logits, const float *probs, const float *seeds,
                             const float *logq, const double *refined, double *stats, float *grad_logits,
                             float *losses, hipStream_t stream);
int launch_lf_to_planes(int N, int M, const float *in, float *out, int negate, hipStream_t stream);
int launch_planes_to_lf(int N, int M, const float *in, float *out, hipStream_t stream);
int launch_argmax_planes(int N, int M, const float *q, int32_t *lab, hipStream_t stream);
int launch_seed_loss_plain(int B, int C, int HW, const float *p, const float *S, float *loss, float *grad, hipStream_t stream);
int launch_expand_loss(int B, int C, int HW, const float *p, const float *stat, double q_fg, double q_bg, float *loss,
                       float *grad, double *terms, hipStream_t stream);
int launch_confusion(size_t n, const unsigned char *gt, const unsigned char *pred, int nclass, int rule_lt,
                     unsigned long long *hist, hipStream_t stream);
int launch_im2col3x3(const void *in, void *out, int B, int H, int W, int C, int dil, hipStream_t stream);
int launch_col2im3x3(const void *cols, void *out, int B, int H, int W, int C, int dil, hipStream_t stream);
int launch_relu_bwd_bias(const void *g, const void *y, void *gm, float *bias_grad, float *part, int part_blocks,
                         long rows, int C, float scale, hipStream_t stream);
int launch_avgpool3x3_s1(const void *in, void *out, int B, int H, int W, int C, hipStream_t stream);
int launch_bias_grad(const void *g, float *bias_grad, float *part, int part_blocks, long rows, int C, hipStream_t stream);
int launch_heads_fwd(const void *const *x, int nbr, const float *w, const float *bias, float *out, int B, int HW, int K, int O,
                     hipStream_t stream);
bool conv3x3_direct_supported(int cin, int cout);
size_t conv3x3_wgrad_workspace(int B, int H, int W, int cin, int cout);
int launch_conv3x3_wgrad(const void *x, const void *g, void *gw, float *workspace, size_t workspace_bytes, int B, int H, int W,
                         int cin, int cout, hipStream_t stream, int out_f32 = 0);
int launch_conv3x3_direct(const void *x, const void *w, const float *bias, void *y, int B, int H, int W, int cin, int cout,
                          int relu, hipStream_t stream, const void *mask = nullptr, float *colsum = nullptr,
                          void *colsum_ws = nullptr, size_t colsum_ws_bytes = 0);
size_t conv3x3_direct_colsum_workspace(int cout);
bool conv_igemm_supported(int cin, int cout, int k);
int launch_conv_igemm_residual(const void *x, const void *w, const float *bias, const void *res, const void *mask, void *y, int dil, int B,
                               int H, int W, int cin, int cout, int k, int relu, hipStream_t stream);
int launch_conv_igemm_split(const void *x3, const void *w, const float *bias, float *y, int dil, int B, int H, int W, int cin, int cout, int k,
                            int relu, hipStream_t stream);
int launch_conv_igemm(const void *const *x, const void *const *w, const float *const *bias, void *const *y, const int *dil,
                      int ngroups, int B, int H, int W, int cin, int cout, int k, int relu, float drop_p, unsigned long long seed,
                      void *workspace, size_t workspace_bytes, hipStream_t stream, const void *const *mask = nullptr,
                      float out_scale = 1.0f, float *const *colsum = nullptr, void *colsum_ws = nullptr, size_t colsum_ws_bytes = 0);
size_t conv_igemm_colsum_workspace(int ngroups, int B, int H, int W, int cout);
size_t conv_igemm_workspace();
int conv_igemm_workspace_status(const void *workspace, hipStream_t stream, int *status);
size_t conv_igemm_wgrad_workspace(int ngroups, int B, int H, int W, int cin, int cout, int k);
int launch_conv_igemm_wgrad(const void *const *x, const void *const *g, void *const *gw, const int *dil, int ngroups, void *workspace,
                            size_t workspace_bytes, int B, int H, int W, int cin, int cout, int k, int out_bf16, hipStream_t stream);
int launch_pack_conv_weight(const float *w, void *fwd, void *dgrad, int cout, int cin, int k, hipStream_t stream, int plain = 0,
                            const float *scale = nullptr);
int launch_conv_igemm_backward_residual(const void *g, const void *wd, const void *x, const void *mask, const void *res, void *gx, void *gw,
                                        const float *gw_scale, int dil, void *wgrad_ws, size_t wgrad_ws_bytes, int B, int H, int W, int cin,
                                        int cout, int k, hipStream_t stream);
int heads_bwd_chunks(int M);
int launch_heads_bwd(const void *const *x, int nbr, const float *w, const float *g, void *gx, size_t gx_branch_stride,
                     float *gw, float *partial, int B, int HW, int K, int O, hipStream_t stream, float relu_scale = 0.0f,
                     float *bias_grad = nullptr, void *colsum_ws = nullptr, size_t colsum_ws_bytes = 0);
size_t heads_bwd_relu_workspace(int nbr, int M, int K);
int launch_add_relu(const void *a, const void *b, void *y, size_t n, hipStream_t stream);
int launch_aspp_shift_sum(const void *yp, const float *bias, float *out, const int *offsets, int J, int O, int CT, int B, int H, int W,
                          hipStream_t stream);
int launch_aspp_shift_gather(const float *g, void *gp, const int *offsets, int J, int O, int CT, int B, int H, int W, hipStream_t stream);
int launch_relu_mask(const void *g, const void *g2, const void *y, void *gm, size_t n, hipStream_t stream);
int launch_igemm_colsum(const float *const *parts, float *const *outs, int ngroups, int rows, int cout, hipStream_t stream);
int launch_maxpool3x3_bwd_relu(const void *gout, const void *code, const void *y, void *gin, float *bias_grad, float *part,
                                int part_blocks, int B, int H, int W, int OH, int OW, int C, hipStream_t stream);
int launch_maxpool3x3_fwd(const void *in, void *out, void *code, int B, int H, int W, int OH, int OW, int C, int stride,
                          hipStream_t stream, bool relu_in = false);
int launch_maxpool3x3_bwd(const void *gout, const void *code, void *gin, int B, int H, int W, int OH, int OW, int C,
                          int stride, hipStream_t stream);

#ifdef __HIPCC__
// exp for the softmax-type normalisations (expAndNormalize densecrf.cpp:98-106, SoftmaxLayer pylayers.py:30-41): the fp64
// exp rounded once to fp32, i.e. the correctly rounded fp32 result (up to double rounding) that a good host libm returns.
// Mean-field amplifies last-bit differences of this exp at pixels where two labels compete, so the device's 1-ulp expf
// would be the largest source of divergence from the CPU path.
__device__ __forceinline__ float exp_cr(float x) { return (float)exp((double)x); }

// SRSRC buffer loads: 32-bit per-lane byte offset + scalar offset against a wave-uniform descriptor
// (one VGPR of address per load in flight instead of two; out-of-range offsets read 0, so index
// loads need no clamping).
using rsrc_t = decltype(__builtin_amdgcn_make_buffer_rsrc((void *)nullptr, (short)0, 0, 0));
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, size_t bytes) {
    // make wave-uniformity provable to the compiler (else every buffer op gets a waterfall loop)
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    const uint32_t nb = __builtin_amdgcn_readfirstlane((uint32_t)bytes);
    void *q = reinterpret_cast<void *>(((uint64_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, (short)0, (int)nb, 0x00020000);
}
__device__ __forceinline__ uint32_t ld_u32(rsrc_t r, uint32_t voff, uint32_t soff = 0) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
}
__device__ __forceinline__ float ld_f32(rsrc_t r, uint32_t voff, uint32_t soff = 0) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ uint32_t ld_u16(rsrc_t r, uint32_t voff, uint32_t soff = 0) {
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0);
}
// Copy nbytes (a multiple of 16 readable at src, 16-byte aligned on both sides) into LDS with a 1024-thread workgroup: U
// 16-byte loads per thread and round, all in flight before the first LDS store of the round.
template <int U>
__device__ __forceinline__ void stage16_to_lds(unsigned char *dst, const void *src, uint32_t nbytes, int tid) {
    using v4 = decltype(__builtin_amdgcn_raw_buffer_load_b128(rsrc_t(), 0, 0, 0));
    const rsrc_t r = make_rsrc(src, nbytes);
    for (uint32_t base = 0; base < nbytes; base += 1024u * 16u * U) {
        v4 x[U];
#pragma unroll
        for (int u = 0; u < U; u++) x[u] = __builtin_amdgcn_raw_buffer_load_b128(r, (uint32_t)tid * 16u, base + (uint32_t)u * (1024u * 16u), 0);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t off = base + (uint32_t)u * (1024u * 16u) + (uint32_t)tid * 16u;
            if (off < nbytes) *reinterpret_cast<v4 *>(dst + off) = x[u];
        }
    }
}

#endif

// ---- seeded region growing ---------------------------------------------------------------
// code: device scratch of B*H*W uint16 (per-pixel classification handed from the first kernel to the second)
int launch_srg(int B, int C, int H, int W, const float *labels, const float *cues, const double *refined,
               double th1, double th2, float *seeds, uint16_t *code, hipStream_t stream);

}  // namespace dsrg
