// Seeded region growing on gfx950: a pixel-parallel classification pass over the
// whole batch, then one workgroup per image that grows the components with the
// label map and growth front resident in LDS (one-pixel halo).
//
// Replaces generate_seed_step (pylayers/pylayers/pylayers.py:237-275) including
// the per-class connected-component labelling it calls
// (pylayers/pylayers/CC_labeling_8.py:112-197).  The reference labels the
// components of every class mask and keeps the ones that contain a cue; because a
// pixel belongs to exactly one class mask (its label-map value), this equals one
// simultaneous flood fill from the cue pixels through equal-label 8-neighbours,
// followed by the reference's exclusion rule — integer work, bit-exact.
#include <math.h>
#include <type_traits>
#include "common.h"

namespace dsrg {

constexpr int kSrgWG = 1024;

// 128-bit row masks (maps up to 128 pixels wide): bit x of row y
struct Mask128 { unsigned long long lo, hi; };
__device__ __forceinline__ Mask128 m_or(Mask128 a, Mask128 b) { return {a.lo | b.lo, a.hi | b.hi}; }
__device__ __forceinline__ Mask128 m_and(Mask128 a, Mask128 b) { return {a.lo & b.lo, a.hi & b.hi}; }
__device__ __forceinline__ Mask128 m_xor(Mask128 a, Mask128 b) { return {a.lo ^ b.lo, a.hi ^ b.hi}; }
__device__ __forceinline__ bool m_eq(Mask128 a, Mask128 b) { return a.lo == b.lo && a.hi == b.hi; }
__device__ __forceinline__ Mask128 m_add(Mask128 a, Mask128 b) {
    Mask128 r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + (r.lo < a.lo ? 1ull : 0ull);
    return r;
}
__device__ __forceinline__ Mask128 m_shl1(Mask128 a) { return {a.lo << 1, (a.hi << 1) | (a.lo >> 63)}; }
__device__ __forceinline__ Mask128 m_shr1(Mask128 a) { return {(a.lo >> 1) | (a.hi << 63), a.hi >> 1}; }
__device__ __forceinline__ Mask128 m_rev(Mask128 a) { return {__brevll(a.hi), __brevll(a.lo)}; }
// all bits of the runs of `m` that contain a bit of `s` (s subset of m): the carry of m + s runs
// from each seed to the end of its run; the mirrored operation covers the other direction
__device__ __forceinline__ Mask128 m_fill_up(Mask128 m, Mask128 s) { return m_or(m_and(m_xor(m_add(m, s), m), m), s); }
__device__ __forceinline__ Mask128 m_fill(Mask128 m, Mask128 s) {
    return m_or(m_fill_up(m, s), m_rev(m_fill_up(m_rev(m), m_rev(s))));
}
__device__ __forceinline__ Mask128 m_dilate3(Mask128 a) { return m_or(a, m_or(m_shl1(a), m_shr1(a))); }
// maps up to 64 pixels wide (the training sizes): one word per row
struct Mask64 { unsigned long long lo; };
__device__ __forceinline__ Mask64 m_or(Mask64 a, Mask64 b) { return {a.lo | b.lo}; }
__device__ __forceinline__ Mask64 m_and(Mask64 a, Mask64 b) { return {a.lo & b.lo}; }
__device__ __forceinline__ bool m_eq(Mask64 a, Mask64 b) { return a.lo == b.lo; }
__device__ __forceinline__ Mask64 m_fill_up(Mask64 m, Mask64 s) { return {(((m.lo + s.lo) ^ m.lo) & m.lo) | s.lo}; }
__device__ __forceinline__ Mask64 m_fill(Mask64 m, Mask64 s) {
    const unsigned long long mr = __brevll(m.lo), sr = __brevll(s.lo);
    return {m_fill_up(m, s).lo | __brevll((((mr + sr) ^ mr) & mr) | sr)};
}
__device__ __forceinline__ Mask64 m_dilate3(Mask64 a) { return {a.lo | (a.lo << 1) | (a.lo >> 1)}; }
__device__ __forceinline__ Mask64 m_shfl(Mask64 a, int src_lane) { return {(unsigned long long)__shfl((long long)a.lo, src_lane, 64)}; }
__device__ __forceinline__ Mask64 m_zero(Mask64) { return {0ull}; }
__device__ __forceinline__ Mask128 m_zero(Mask128) { return {0ull, 0ull}; }
__device__ __forceinline__ Mask64 m_from(Mask128 a, Mask64) { return {a.lo}; }
__device__ __forceinline__ Mask128 m_from(Mask128 a, Mask128) { return a; }
__device__ __forceinline__ Mask128 m_to128(Mask64 a) { return {a.lo, 0ull}; }
__device__ __forceinline__ Mask128 m_to128(Mask128 a) { return a; }

__device__ __forceinline__ Mask128 m_shfl(Mask128 a, int src_lane) {
    Mask128 r;
    r.lo = (unsigned long long)__shfl((long long)a.lo, src_lane, 64);
    r.hi = (unsigned long long)__shfl((long long)a.hi, src_lane, 64);
    return r;
}

// ---- stage 1: classification, pixel-parallel over the whole batch (grid = pixel tiles x images) -------------------------
// Per pixel one sweep over the labels: highest cued class (numpy's fancy assignment, pylayers.py:248-250), cue sum, argmax /
// max of the float64 marginals over the PRESENT classes with the first maximum winning (pylayers.py:241-243), the threshold
// rule (pylayers.py:251-257) -> one 16-bit code per pixel for the growth stage, and the pass-through copy seeds = cues
// (absent classes and non-grown pixels keep their cues, pylayers.py:259-273).
//   code bits 0..7: label-map value (class + 1; 0 = not part of any present class's mask), bit 8: the pixel is a cue of its
//   own class (a seed of its component, :266), bit 9: exclusion rule (own cue absent and exactly one other cue, :268-269)
constexpr int kSrgCodeSeed = 1 << 8, kSrgCodeExcl = 1 << 9;
constexpr int kSrgTile = 256;
template <int CT>   // CT > 0: C <= CT, every load of a pixel issued before the first use; CT = 0: any C, 8 labels per batch
__global__ __launch_bounds__(kSrgTile) void srg_classify_kernel(int C, int N, const float *__restrict__ labels,
                                                                 const float *__restrict__ cues,
                                                                 const double *__restrict__ refined, double th1, double th2,
                                                                 float *__restrict__ seeds, uint16_t *__restrict__ code) {
    const int b = blockIdx.y, p = blockIdx.x * kSrgTile + threadIdx.x;
    const float *lab = labels + (size_t)b * C;           // workgroup-uniform: scalar loads
    const float *cu = cues + (size_t)b * C * N;
    const double *rf = refined + (size_t)b * C * N;
    float *out = seeds + (size_t)b * C * N;
    const int pc = min(p, N - 1);
    int lmv = 0, best = -1;
    float cuesum = 0.0f, own_best = 0.0f;                // own_best: the cue of the arg-max class at this pixel
    double v = 0.0;
    constexpr int CH = CT > 0 ? CT : 8;
    float own_lm = 0.0f;                                 // the cue of the highest cued class (1 by construction when lmv > 0)
    for (int c0 = 0; c0 < C; c0 += CH) {
        float sc[CH];
        double rc[CH];
#pragma unroll
        for (int q = 0; q < CH; q++) {
            const size_t o = (size_t)min(c0 + q, C - 1) * N + pc;
            sc[q] = cu[o];
            rc[q] = rf[o];
        }
#pragma unroll
        for (int q = 0; q < CH; q++) {
            const int c = c0 + q;
            if (c < C) {
                if (p < N) out[(size_t)c * N + p] = sc[q];                          // seeds start as the cues
                if (sc[q] > 0.0f) { lmv = c + 1; own_lm = sc[q]; }
                cuesum += sc[q];
                if (lab[c] == 1.0f && (best < 0 || rc[q] > v)) { v = rc[q]; best = c; own_best = sc[q]; }
            }
        }
    }
    if (p >= N) return;
    float own = own_lm;
    if (best >= 0 && v > th2) {                          // pylayers.py:253-257, strict float64 compares
        if (best != 0) { lmv = best + 1; own = own_best; }
        else if (v > th1) { lmv = 1; own = own_best; }
    }
    const bool active = lmv > 0 && lab[lmv - 1] == 1.0f;                          // only present classes are grown (:259)
    int cd = 0;
    if (active) {
        cd = lmv;
        if (own == 1.0f) cd |= kSrgCodeSeed;                                        // seeds of the component (:266)
        else if (cuesum == 1.0f) cd |= kSrgCodeExcl;                                // cued by exactly one OTHER class (:268-269)
    }
    code[(size_t)b * N + p] = (uint16_t)cd;
}

// ---- stage 2: growth, one workgroup per image ----------------------------------------------------------------------------
__global__ __launch_bounds__(kSrgWG) void srg_grow_kernel(int C, int H, int W, const uint16_t *__restrict__ code,
                                                          float *__restrict__ seeds, int mask_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int N = H * W, Wp = W + 2, Np = (H + 2) * Wp;
    unsigned char *lm = smem;                 // [(H+2)*(W+2)] label-map value (class+1), 0 = none / halo
    unsigned char *grown = smem + ((Np + 15) & ~15);      // same shape: 1 = member of a seeded component
    unsigned char *excl = grown + ((Np + 15) & ~15);      // [N] 1 = the pixel's own plane is not to be written (seed already / rule)
    uint32_t *present = reinterpret_cast<uint32_t *>(excl + ((N + 15) & ~15));    // [4] labels that occur in the label map
    const uint16_t *cd = code + (size_t)b * N;
    float *out = seeds + (size_t)b * C * N;

    for (int p = tid; p < Np; p += kSrgWG) { lm[p] = 0; grown[p] = 0; }
    if (tid < 4) present[tid] = 0u;
    __syncthreads();
    for (int p = tid; p < N; p += kSrgWG) {
        const int k = cd[p];
        const int y = p / W, x = p - y * W;
        const int pp = (y + 1) * Wp + (x + 1);
        const int l = k & 0xff;
        lm[pp] = (unsigned char)l;
        grown[pp] = (k & kSrgCodeSeed) ? 1 : 0;
        excl[p] = (k & (kSrgCodeSeed | kSrgCodeExcl)) ? 1 : 0;     // a seed's plane already holds 1 (its cue)
        if (l) atomicOr(&present[(l - 1) >> 5], 1u << ((l - 1) & 31));
    }
    __syncthreads();

    // ---- grow to the fixed point.  A pixel joins when an 8-neighbour with the same label is a member.
    if (W <= 128 && H <= 128 && (size_t)2 * (C + 1) * H * sizeof(Mask128) <= (size_t)mask_bytes) {
        // Row-mask formulation: per label l and row y, M[l][y] = pixels carrying l, G[l][y] = members.
        // Masks are assembled with wave ballots; then ONE lane per label sweeps the rows down and up —
        // a row step is "dilate the neighbouring row's members by one pixel, intersect with M, close
        // along the row's runs with a carry chain" — until nothing changes (a handful of sweeps).
        Mask128 *Mm = reinterpret_cast<Mask128 *>(reinterpret_cast<unsigned char *>(present) + 16);       // [(C+1)][H]
        Mask128 *Gm = Mm + (size_t)(C + 1) * H;
        const int wave = tid >> 6, nwaves = kSrgWG >> 6;
        for (int y = wave; y < H; y += nwaves) {
            for (int xh = 0; xh < W; xh += 64) {
                const int x = xh + lane;
                const int pp = (y + 1) * Wp + (x + 1);
                const unsigned char l = (x < W) ? lm[pp] : 0;
                const unsigned char gr = (x < W) ? grown[pp] : 0;
                for (int c = 1; c <= C; c++) {
                    if (!((present[(c - 1) >> 5] >> ((c - 1) & 31)) & 1u)) continue;   // workgroup-uniform
                    const unsigned long long mb = __ballot(l == c), gb = __ballot(l == c && gr);
                    if (lane == 0) {
                        if (xh == 0) { Mm[c * H + y].lo = mb; Gm[c * H + y].lo = gb; if (W <= 64) { Mm[c * H + y].hi = 0; Gm[c * H + y].hi = 0; } }
                        else { Mm[c * H + y].hi = mb; Gm[c * H + y].hi = gb; }
                    }
                }
            }
        }
        __syncthreads();
        // one wave per present label, one lane per row (rows y = lane + 64k): every iteration each row
        // absorbs its two neighbouring rows' members (dilated by one pixel, i.e. incl. diagonals),
        // restricted to its own label mask and closed along the row's runs; neighbours travel by
        // lane shuffles, state stays in registers, convergence is a wave vote — no barrier inside.
        // A component of height h needs ~h iterations, so an iteration is kept short: one 64-bit word per row and one row
        // per lane when the map allows (41x41: a quarter of the 128x128 form's work per iteration)
        auto grow = [&](auto mask_tag, auto rpl_tag) {
            using M_t = decltype(mask_tag);
            constexpr int RPL = decltype(rpl_tag)::value;
            for (int c = 1 + wave; c <= C; c += nwaves) {
                if (!((present[(c - 1) >> 5] >> ((c - 1) & 31)) & 1u)) continue;       // workgroup-uniform
                M_t M[RPL], G[RPL];
#pragma unroll
                for (int k = 0; k < RPL; k++) {
                    const int y = lane + 64 * k;
                    M[k] = (y < H) ? m_from(Mm[(size_t)c * H + y], M_t{}) : m_zero(M_t{});
                    G[k] = (y < H) ? m_fill(M[k], m_from(Gm[(size_t)c * H + y], M_t{})) : m_zero(M_t{});
                }
                for (;;) {
                    bool changed = false;
                    M_t up[RPL], dn[RPL];                                          // members of rows y-1 and y+1
#pragma unroll
                    for (int k = 0; k < RPL; k++) {
                        up[k] = m_shfl(G[k], (lane + 63) & 63);                    // from lane-1 (row y-1), wraps
                        dn[k] = m_shfl(G[k], (lane + 1) & 63);                     // from lane+1 (row y+1), wraps
                    }
                    // lane 0's row above is row 64k-1 = lane 63 of chunk k-1; lane 63's row below is lane 0 of chunk k+1
#pragma unroll
                    for (int k = RPL - 1; k >= 0; k--) {
                        if (lane == 0) up[k] = (k > 0) ? up[k - 1] : m_zero(M_t{});
                    }
#pragma unroll
                    for (int k = 0; k < RPL; k++) {
                        if (lane == 63) dn[k] = (k + 1 < RPL) ? dn[k + 1] : m_zero(M_t{});
                    }
#pragma unroll
                    for (int k = 0; k < RPL; k++) {
                        const M_t add = m_and(m_dilate3(m_or(up[k], dn[k])), M[k]);
                        const M_t g1 = m_fill(M[k], m_or(G[k], add));
                        if (!m_eq(g1, G[k])) { G[k] = g1; changed = true; }
                    }
                    if (!__any(changed)) break;
                }
#pragma unroll
                for (int k = 0; k < RPL; k++) {
                    const int y = lane + 64 * k;
                    if (y < H) Gm[(size_t)c * H + y] = m_to128(G[k]);
                }
            }
        };
        if (W <= 64 && H <= 64) grow(Mask64{}, std::integral_constant<int, 1>{});
        else if (W <= 64) grow(Mask64{}, std::integral_constant<int, 2>{});
        else if (H <= 64) grow(Mask128{}, std::integral_constant<int, 1>{});
        else grow(Mask128{}, std::integral_constant<int, 2>{});
        __syncthreads();
        for (int p = tid; p < N; p += kSrgWG) {
            const int y = p / W, x = p - y * W;
            const int pp = (y + 1) * Wp + (x + 1);
            const int l = lm[pp];
            if (l != 0) {
                const Mask128 g = Gm[(size_t)l * H + y];
                grown[pp] = (unsigned char)(((x < 64 ? g.lo >> x : g.hi >> (x - 64)) & 1ull));
            }
        }
        __syncthreads();
    } else {
        for (;;) {
            int changed = 0;
            for (int p = tid; p < N; p += kSrgWG) {
                const int y = p / W, x = p - y * W;
                const int pp = (y + 1) * Wp + (x + 1);
                const unsigned char l = lm[pp];
                if (l != 0 && grown[pp] == 0) {
                    const int up = pp - Wp, dn = pp + Wp;
                    const bool hit = (grown[up - 1] && lm[up - 1] == l) || (grown[up] && lm[up] == l) ||
                                     (grown[up + 1] && lm[up + 1] == l) || (grown[pp - 1] && lm[pp - 1] == l) ||
                                     (grown[pp + 1] && lm[pp + 1] == l) || (grown[dn - 1] && lm[dn - 1] == l) ||
                                     (grown[dn] && lm[dn] == l) || (grown[dn + 1] && lm[dn + 1] == l);
                    if (hit) { grown[pp] = 1; changed = 1; }
                }
            }
            // frontier empty?  64-bit wave ballot, then OR across the workgroup's waves
            const int wave_changed = __ballot(changed) != 0ull;
            if (!__syncthreads_or(wave_changed)) break;
        }
    }

    // every member pixel of a seeded component joins its class's seeds unless excluded (pylayers.py:271-273); all other
    // entries keep the cues the classification stage copied
    for (int p = tid; p < N; p += kSrgWG) {
        const int y = p / W, x = p - y * W;
        const int pp = (y + 1) * Wp + (x + 1);
        const int cls = (int)lm[pp] - 1;
        if (cls >= 0 && grown[pp] && !excl[p]) out[(size_t)cls * N + p] = 1.0f;
    }
}

int launch_srg(int B, int C, int H, int W, const float *labels, const float *cues, const double *refined,
               double th1, double th2, float *seeds, uint16_t *code, hipStream_t stream) {
    if (C < 1 || C > kMaxLabels) return set_error(DSRG_ERR_UNSUPPORTED, "SRG supports 1..%d classes, got %d", kMaxLabels, C);
    const int N = H * W;
    const size_t Np = (size_t)(H + 2) * (W + 2);
    size_t lds = 2 * ((Np + 15) & ~(size_t)15) + (((size_t)N + 15) & ~(size_t)15) + 16;
    if (lds > 150 * 1024) return set_error(DSRG_ERR_UNSUPPORTED, "SRG map %dx%d exceeds LDS", H, W);
    // row masks for the fast growth path: 2 x (C+1) x H 128-bit masks
    size_t mask_bytes = (size_t)2 * (C + 1) * H * sizeof(Mask128);
    if (W > 128 || H > 128 || lds + mask_bytes > 150 * 1024) mask_bytes = 0;
    lds += mask_bytes;
    static LdsGrant granted;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&srg_grow_kernel), lds, granted);
    if (rc) return rc;
    const dim3 grid((N + kSrgTile - 1) / kSrgTile, B);
    if (C <= 21)
        hipLaunchKernelGGL(srg_classify_kernel<21>, grid, dim3(kSrgTile), 0, stream, C, N, labels, cues, refined, th1, th2,
                           seeds, code);
    else
        hipLaunchKernelGGL(srg_classify_kernel<0>, grid, dim3(kSrgTile), 0, stream, C, N, labels, cues, refined, th1, th2,
                           seeds, code);
    DSRG_LAUNCH_CHECK();
    hipLaunchKernelGGL(srg_grow_kernel, dim3(B), dim3(kSrgWG), lds, stream, C, H, W, code, seeds, (int)mask_bytes);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

}  // namespace dsrg
