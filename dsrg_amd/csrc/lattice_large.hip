// Dense CRF for maps that do not fit the LDS-resident path (test-time CRF at full image resolution,
// training/tools/test-ms.py:99-106: ~500x375 pixels, scale_factor 1, log-probability unaries).
//
// Same arithmetic as lattice.hip / meanfield.hip (shared embedding in embed.h, same accumulation
// orders), but every structure lives in HBM/L2: a global open-addressing hash table, 32-bit vertex
// ids, lattice values as [vertex][CP] rows (CP = labels padded to a multiple of 4 — the layout the
// reference's sseCompute uses, permutohedral.cpp:531-535) moved 16 bytes per lane.  This is the regime
// where the filter really is bandwidth-bound (values: M x CP x 4 B per buffer, streamed 2(d+1)+2 times).
//
// Build-time sorting/scanning uses hipCUB (DeviceRadixSort / DeviceScan): plumbing, once per image.
#include <math.h>
#include <cstring>
#include "common.h"
#include <cstdlib>
#include "embed.h"

namespace dsrg {

constexpr uint32_t kEmptyL = 0xFFFFFFFFu;

struct LargeLattice {
    int d, N, Npad, E, Epad, Mcap, cap;
    // batch mode (nimg > 1, dsrg_crf_create_batch): the object filters nimg same-sized images at once.  Pixel index space =
    // nimg slots of Npimg = 4 * ceil(Nimg / 4) pixels: image b's Nimg pixels, then the up to three zero-feature pixels with
    // which the reference's SSE loop pads ITS image (permutohedral.cpp:191-199) — here ordinary pixels with barycentric weight
    // 0.  Every key carries its image's number (embedding kernel), so images share no vertex and the build, the splat rows
    // (an image's entries in the reference's order, its padding entries last with weight 0) and the blur neighbours come out
    // per image exactly as in a single-image object; every kernel below just sees one lattice over N = nimg * Npimg pixels.
    int nimg, Nimg, Npimg;
    int M_host;
    uint32_t *key_e, *table, *slot_e, *first, *scanned, *key_v, *vid, *nb1, *nb2;
    uint32_t *ent_vid, *ent_idx, *srt_vid, *srt_idx, *cnt, *row_start, *csr_pix;
    float *bary, *csr_w, *norm;
    int *M;                  // [0] vertex count, [1] number of splat segments T, [2] number of multi-segment vertices
    // splat work units: a vertex's row of entries is cut into segments of at most kSplatSeg entries (rows are heavy-tailed at
    // image resolution: median 13 entries, maximum ~400 — one lane group per ROW made the longest row the kernel's duration)
    uint32_t *seg_start;     // [M+1] first segment of vertex v (exclusive scan of max(1, ceil(len / kSplatSeg)))   (aliases `first`)
    uint32_t *seg_cnt;       // [M+1] scan input                                                                (aliases `scanned`)
    uint32_t *seg_v;         // [T]   vertex of segment s                                                      (aliases `slot_e`)
    uint32_t *multi_v;       // [..]  vertices with more than one segment, any order                           (aliases `key_e`)
    int T_host, nmulti_host;
    int seg_len;             // entries per splat segment
};
// (4 096 since round 6: a vertex's row is summed WHOLE, entry by entry in the reference's order, unless it is longer than any row of a
// natural image — a flat region of more than ~4 000 pixels inside one 80-pixel lattice cell.  With 64-entry segments the partial
// sums' reassociation, amplified by ten softmax iterations at weight 10, put the worst sweep case at 7.4e-5 of the 1e-4 contract
// (138 x 163, dark corner); whole rows: 4.4e-6, at the same 1.44 ms per natural 321 x 321 image and 0.80 ms batched.  A flat image
// pays: 2.7 -> 5.0 ms.  profiles/r06_splat_segments.txt; DSRG_SPLAT_SEG overrides, tools only)
constexpr int kSplatSeg = 4096;
constexpr int kLargeBatchMax = 8;          // images per batched object: the image number's room in the d = 2 keys (lg_embed_kernel)


// ---------------------------------------------------------------------------------------------
template <int D>
__global__ void lg_embed_kernel(LargeLattice L, LatticeFeat F, const unsigned char *__restrict__ im) {
    constexpr int D1 = D + 1, KW = KeyWords<D>::value;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.Npad) return;
    uint32_t keys[D1][KW];
    float bc[D1];
    if (L.nimg > 1) {
        const int b = i / L.Npimg, il = i - b * L.Npimg;
        const bool real = il < L.Nimg;
        float pr = 0.0f, pg = 0.0f, pb = 0.0f;
        if (D == 5 && real) {
            const unsigned char *px = im + ((size_t)b * L.Nimg + il) * 3;
            pr = (float)px[0]; pg = (float)px[1]; pb = (float)px[2];
        }
        embed_pixel_rgb<D>(F, il, L.Nimg, pr, pg, pb, keys, bc);
        bool out_of_range = false;
#pragma unroll
        for (int r = 0; r < D1; r++) {
            if (!real) bc[r] = 0.0f;                        // a padding pixel only creates its vertices (the reference never splats it)
            if (D == 5) {
                keys[r][KW - 1] |= (uint32_t)b << 16;       // 5 x 16 bits of key in three words: the last word's upper half is free
            } else {
                // two 16-bit fields fill the only word: the image number goes on top of the first field, 4096 per image — room for
                // eight images of coordinates within +-2048 (a 500-pixel side at sigma = 3 reaches ~300); checked, never assumed
                const uint32_t f0 = keys[r][0] & 0xffffu;
                out_of_range |= f0 < 0x8000u - 2040u || f0 >= 0x8000u + 2040u;     // (a neighbour key reaches 3 further)
                keys[r][0] += (uint32_t)b * 4096u;
            }
        }
        if (out_of_range) atomicOr(reinterpret_cast<unsigned int *>(L.M + 3), 1u);
    } else {
        embed_pixel<D>(F, i, L.N, im, keys, bc);
    }
#pragma unroll
    for (int r = 0; r < D1; r++) {
#pragma unroll
        for (int q = 0; q < KW; q++) L.key_e[((size_t)i * D1 + r) * KW + q] = keys[r][q];
        if (i < L.N) L.bary[(size_t)r * L.N + i] = bc[r];
    }
}

// For every lane of a wave: the lowest lane that holds the same value (itself when it is the first).  A wave's 64 consecutive
// entries are ~10 neighbouring pixels whose simplices share most of their vertices, so a wave holds ~10 distinct keys: one
// lane per key then goes to the table instead of 64 contending for the same few slots.  64 steps of readlane + compare.
template <int KW>
__device__ __forceinline__ int wave_first_lane_with_same(const uint32_t (&w)[KW], bool valid) {
    const int lane = threadIdx.x & 63;
    int leader = lane;
    bool found = false;
#pragma unroll
    for (int j = 0; j < 64; j++) {
        bool same = __builtin_amdgcn_readlane((int)valid, j) != 0;
#pragma unroll
        for (int q = 0; q < KW; q++) same = same && ((uint32_t)__builtin_amdgcn_readlane((int)w[q], j) == w[q]);
        if (same && !found && valid) { leader = j; found = true; }
    }
    return leader;
}

// open addressing, linear probing; a slot keeps the smallest entry index of its key (= first occurrence
// in the reference's visiting order, permutohedral.cpp:261-276).  Within a wave the first lane of every distinct key (it
// holds the key's smallest entry index of the wave) inserts; the others take its slot.
template <int D>
__global__ void lg_insert_kernel(LargeLattice L) {
    constexpr int KW = KeyWords<D>::value;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = e < L.Epad;
    uint32_t w[KW];
#pragma unroll
    for (int q = 0; q < KW; q++) w[q] = 0;
    if (valid) load_key<KW>(w, L.key_e + (size_t)e * KW);
    const int lane = threadIdx.x & 63;
    const int leader = wave_first_lane_with_same<KW>(w, valid);
    const uint32_t mask = (uint32_t)L.cap - 1u;
    uint32_t h = hash_key<KW>(w) & mask;
    if (valid && leader == lane) {
        for (;;) {
            const uint32_t old = atomicCAS(&L.table[h], kEmptyL, (uint32_t)e);
            if (old == kEmptyL) break;
            if (key_eq<KW>(w, L.key_e + (size_t)old * KW)) { atomicMin(&L.table[h], (uint32_t)e); break; }
            h = (h + 1) & mask;
        }
    }
    h = (uint32_t)__shfl((int)h, leader, 64);
    if (valid) L.slot_e[e] = h;
}

__global__ void lg_first_kernel(LargeLattice L) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= L.Epad) return;
    L.first[e] = (L.table[L.slot_e[e]] == (uint32_t)e) ? 1u : 0u;
}

template <int D>
__global__ void lg_assign_kernel(LargeLattice L) {
    constexpr int KW = KeyWords<D>::value;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= L.Epad) return;
    if (e == L.Epad - 1) *L.M = (int)(L.scanned[e] + L.first[e]);
    if (L.first[e]) {
        const uint32_t id = L.scanned[e];           // ids in first-occurrence order = the reference's ids
#pragma unroll
        for (int t = 0; t < KW; t++) L.key_v[(size_t)id * KW + t] = L.key_e[(size_t)e * KW + t];
        L.table[L.slot_e[e]] = id;                  // (all reads of the entry indices finished in lg_first_kernel)
    }
}

// vertex id of every entry + row counts of the splat CSR: one atomic per distinct vertex of a wave (the first lane that
// holds it adds the wave's count; ~26 entries share a vertex at 321x321)
__global__ void lg_vid_kernel(LargeLattice L, int D1) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = e < L.E;
    uint32_t v[1] = {0u};
    if (valid) {
        const int i = e / D1, r = e - i * D1;
        v[0] = L.table[L.slot_e[e]];
        L.vid[(size_t)r * L.N + i] = v[0];
        L.ent_vid[e] = v[0];
        L.ent_idx[e] = (uint32_t)e;
    }
    const int lane = threadIdx.x & 63;
    const int leader = wave_first_lane_with_same<1>(v, valid);
    // members of my group: lanes whose leader is me
    uint32_t n = 0;
#pragma unroll
    for (int j = 0; j < 64; j++) n += (__builtin_amdgcn_readlane(valid ? leader : -1, j) == lane) ? 1u : 0u;
    if (valid && leader == lane) atomicAdd(&L.cnt[v[0]], n);
}

// blur neighbours (permutohedral.cpp:303-318); M is the "no neighbour" id (row M of the values is zero)
template <int D>
__global__ void lg_neigh_kernel(LargeLattice L) {
    constexpr int D1 = D + 1, KW = KeyWords<D>::value;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const int M = *L.M;
    if (v >= M) return;
    uint32_t w[KW];
    load_key<KW>(w, L.key_v + (size_t)v * KW);
    const uint32_t mask = (uint32_t)L.cap - 1u;
#pragma unroll
    for (int j = 0; j < D1; j++) {
#pragma unroll
        for (int s = 0; s < 2; s++) {
            uint32_t q[KW];
            neighbour_key<D>(q, w, j, s != 0);
            uint32_t h = hash_key<KW>(q) & mask;
            uint32_t found = (uint32_t)M;
            for (;;) {
                const uint32_t t = L.table[h];
                if (t == kEmptyL) break;
                if (key_eq<KW>(q, L.key_v + (size_t)t * KW)) { found = t; break; }
                h = (h + 1) & mask;
            }
            (s ? L.nb2 : L.nb1)[(size_t)j * L.Mcap + v] = found;
        }
    }
}

__global__ void lg_csr_kernel(LargeLattice L, int D1) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= L.E) return;
    const uint32_t e = L.srt_idx[pos];              // stable sort by vertex: entries of a vertex in visiting order
    const uint32_t i = e / (uint32_t)D1, r = e - i * (uint32_t)D1;
    L.csr_pix[pos] = i;
    L.csr_w[pos] = L.bary[(size_t)r * L.N + i];
}

// ---- splat segments: per vertex max(1, ceil(len / kSplatSeg)) units
__global__ void lg_seg_count_kernel(LargeLattice L) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const int M = *L.M;
    if (v > L.Mcap) return;
    if (v >= M) { L.seg_cnt[v] = 0u; return; }
    const uint32_t len = L.row_start[v + 1] - L.row_start[v];
    L.seg_cnt[v] = len <= (uint32_t)L.seg_len ? 1u : (len + L.seg_len - 1) / L.seg_len;
}
__global__ void lg_seg_fill_kernel(LargeLattice L) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const int M = *L.M;
    if (v == 0) { L.M[1] = (int)L.seg_start[M]; }
    if (v >= M) return;
    const uint32_t s0 = L.seg_start[v], s1 = L.seg_start[v + 1];
    for (uint32_t q = s0; q < s1; q++) L.seg_v[q] = (uint32_t)v;
    if (s1 - s0 > 1u) L.multi_v[atomicAdd(reinterpret_cast<unsigned int *>(&L.M[2]), 1u)] = (uint32_t)v;
}

// ---- one-channel filter with Permutohedral::seqCompute semantics (permutohedral.cpp:476-527) for the
// normalisation vector (pairwise.cpp:44,54-57)
__global__ void lg_splat1_kernel(LargeLattice L, float *__restrict__ val) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const int M = *L.M;
    if (v == 0) val[M] = 0.0f;
    if (v >= M) return;
    // rows are heavy-tailed (up to hundreds of entries): the loads do not depend on the sum, so eight are in flight per step
    // while the additions keep the reference's order
    float s = 0.0f;
    const uint32_t p1 = L.row_start[v + 1];
    uint32_t pos = L.row_start[v];
    for (; pos + 8 <= p1; pos += 8) {
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; u++) w[u] = L.csr_w[pos + u];
#pragma unroll
        for (int u = 0; u < 8; u++) s = s + w[u] * 1.0f;
    }
    for (; pos < p1; pos++) s = s + L.csr_w[pos] * 1.0f;
    val[v] = s;
}
__global__ void lg_blur1_kernel(LargeLattice L, int j, const float *__restrict__ a, float *__restrict__ b) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const int M = *L.M;
    if (v == 0) b[M] = 0.0f;
    if (v >= M) return;
    const float s = a[L.nb1[(size_t)j * L.Mcap + v]] + a[L.nb2[(size_t)j * L.Mcap + v]];
    b[v] = (float)((double)a[v] + 0.5 * (double)s);
}
__global__ void lg_norm_kernel(LargeLattice L, int D1, const float *__restrict__ val) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.N) return;
    const float alpha = 1.0f / (1.0f + exp2f(-(float)(D1 - 1)));
    float out = 0.0f;
    for (int r = 0; r < D1; r++) {
        float t = L.bary[(size_t)r * L.N + i] * val[L.vid[(size_t)r * L.N + i]];
        t = t * alpha;
        out = out + t;
    }
    L.norm[i] = (float)(1.0 / sqrt((double)out + 1e-20));
}

// ---- CP-channel filter (Permutohedral::sseCompute, permutohedral.cpp:529-589); in/out are [N][CP]
// one wave per vertex, lanes = channels: ordered accumulation of the vertex's row of (pixel, weight) entries
// the splat's input is Q * norm (pairwise.cpp:66), stored per lattice by the update kernels.  Measured and dropped in round 6 (experiment
// build, make EXP=2 EXPSRC=lattice_large exp): forming it inside the splat from Q and the pixel's norm (the same fp32 product; two
// 96-byte row writes per pixel and iteration less, both lattices gather one array) — the splat is a chain of dependent gathers and the
// extra norm load sits in it: 169 -> 229 us per batch-8 launch, 0.772 -> 0.807 ms per image (profiles/r06_crf_batch8_ab.txt)
#if defined(DSRG_EXP) && (DSRG_EXP & 2)
constexpr bool kSplatNormOnTheFly = true;
#else
constexpr bool kSplatNormOnTheFly = false;
#endif
__global__ __launch_bounds__(256) void lg_update_kernel(int N, int C, int CP, const float *__restrict__ neg_unary,
                                                        const float *__restrict__ t_g, const float *__restrict__ t_b,
                                                        int use_msgs, float *__restrict__ q, const float *__restrict__ norm_b,
                                                        const float *__restrict__ norm_g, float *__restrict__ qn_b,
                                                        float *__restrict__ qn_g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *row = reinterpret_cast<float *>(smem);            // [256][CP+1]
    const int P = CP + 1;
    const size_t i0 = (size_t)blockIdx.x * 256;
    const int npix = (int)min((size_t)256, (size_t)N - i0);
    const int nelem = npix * CP;
    for (int k = threadIdx.x; k < nelem; k += 256) {
        const size_t o = i0 * CP + k;
        float v = neg_unary[o];                               // tmp1 = -unary
        if (use_msgs) { v = v - t_g[o]; v = v - t_b[o]; }     // tmp1 -= tmp2 (Gaussian, then bilateral)
        row[(k / CP) * P + (k % CP)] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < npix) {
        float *r = row + threadIdx.x * P;
        float mx = -INFINITY;
        for (int c = 0; c < C; c++) mx = fmaxf(mx, r[c]);
        float sum = 0.0f;
        for (int c = 0; c < C; c++) { const float e = exp_cr(r[c] - mx); r[c] = e; sum = sum + e; }
        for (int c = 0; c < C; c++) r[c] = r[c] / sum;
        for (int c = C; c < CP; c++) r[c] = 0.0f;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nelem; k += 256) {
        const float v = row[(k / CP) * P + (k % CP)];
        q[i0 * CP + k] = v;
        // the splat's input, in = Q * norm (pairwise.cpp:66), formed once per pixel here instead of once per gather there
        if (!kSplatNormOnTheFly) {
            qn_b[i0 * CP + k] = v * norm_b[i0 + k / CP];
            qn_g[i0 * CP + k] = v * norm_g[i0 + k / CP];
        }
    }
}
// label-fastest [N][C] (host layout of DenseCRFWrapper) <-> padded rows [N][CP]
// XCD-aware work map.  Consecutive workgroups go to the 8 XCDs round-robin and each XCD has its own 4 MB L2 (cold at every
// kernel start), so work item w of `total` is handed to the workgroup whose XCD owns the contiguous eighth that contains it:
// vertex ids are first-occurrence order = image order, so an eighth of the vertices (or pixels) is an image strip, and the
// ~9 entries that gather a pixel's row (or the ~30 pixels that gather a vertex's row) meet in ONE L2 instead of eight.
// Returns the first unit of block `b` (units per block `per_block`), or -1 when the block is padding.
__device__ __forceinline__ long long xcd_strip_first(unsigned b, size_t total, unsigned per_block, size_t *strip_end) {
    const unsigned x = b & 7u, s = b >> 3;
    const size_t per_xcd = ((total + 7) / 8 + per_block - 1) / per_block * per_block;    // units per XCD, whole blocks
    const size_t first = (size_t)x * per_xcd + (size_t)s * per_block;
    const size_t end = min((size_t)(x + 1) * per_xcd, total);
    *strip_end = end;
    return first < end ? (long long)first : -1;
}
__host__ __device__ static inline unsigned xcd_strip_blocks(size_t total, unsigned per_block) {
    const size_t per_xcd = ((total + 7) / 8 + per_block - 1) / per_block;               // blocks per XCD
    return (unsigned)(8 * per_xcd);
}

// ---- both lattices in one launch (the mean-field loop is a chain of short dependent launches: 14 -> 8 per iteration) ----
// Splat: ordered sum of weight * (in * norm) over the vertex's entry list (pairwise.cpp:66, permutohedral.cpp:529-545);
// blur: x + 0.5 (n1 + n2) per axis, Jacobi (:547-565); slice: barycentric gather * alpha, * norm, * -w (:567-589,
// pairwise.cpp:72-79); update: -unary - gaussian - bilateral, expAndNormalize (densecrf.cpp:98-131).
// One segment of vertex v's row: the ordered sum of weight * in over entries [p0, p1), in = Q * norm — at most kSplatSeg of them.
// A row is a chain of dependent gathers (entry -> pixel -> value): eight entries per round, the (pixel, weight) words of the
// NEXT round fetched while this round's gathers are in flight (a round costs one memory round trip, not two); the tail rides
// in the same 8-wide code with weight 0 (s + 0 * x = s exactly).
__device__ __forceinline__ float lg_splat_segment(const LargeLattice &L, uint32_t p0, uint32_t p1, int lane, int CP,
                                                  const float *__restrict__ in) {
    float s = 0.0f;
    constexpr int U = 8;
    uint32_t px[U];
    float w[U];
    auto load_idx = [&](uint32_t pos, uint32_t (&px_)[U], float (&w_)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t q = min(pos + (uint32_t)u, p1 - 1u);          // clamped: always a valid entry of this segment
            px_[u] = L.csr_pix[q];
            const float wq = L.csr_w[q];
            w_[u] = (pos + (uint32_t)u < p1) ? wq : 0.0f;
        }
    };
    if (p1 > p0) load_idx(p0, px, w);
    for (uint32_t pos = p0; pos < p1; pos += U) {
        float x[U];
#pragma unroll
        for (int u = 0; u < U; u++) x[u] = in[(size_t)px[u] * CP + lane];
        if (kSplatNormOnTheFly) {                            // in = Q: times the pixel's norm here (the same fp32 product the update kernel
            float nm[U];                                     // used to store — two row writes and a second 79 MB input per iteration less)
#pragma unroll
            for (int u = 0; u < U; u++) nm[u] = L.norm[px[u]];
#pragma unroll
            for (int u = 0; u < U; u++) x[u] = x[u] * nm[u];
        }
        uint32_t npx[U];
        float nw[U];
        const bool more = pos + U < p1;
        if (more) load_idx(pos + U, npx, nw);
#pragma unroll
        for (int u = 0; u < U; u++) s = s + w[u] * x[u];
        if (more) {
#pragma unroll
            for (int u = 0; u < U; u++) { px[u] = npx[u]; w[u] = nw[u]; }
        }
    }
    return s;
}
// unit `u` of a lattice's splat: segment u < T, or the zero row (u == T).  A vertex with ONE segment (rows of up to kSplatSeg
// entries: the reference's accumulation order, exactly) writes its value; longer rows write per-segment partial sums that
// lg_combine_kernel adds up in segment order (deterministic; differs from the strictly sequential sum by reassociation only).
__device__ __forceinline__ void lg_splat_unit(const LargeLattice &L, size_t u, int lane, int CP, const float *__restrict__ in,
                                              float *__restrict__ val, float *__restrict__ part) {
    const int M = L.M[0], T = L.M[1];
    if (u == (size_t)T) { val[(size_t)M * CP + lane] = 0.0f; return; }
    const uint32_t v = L.seg_v[u];
    const uint32_t s0 = L.seg_start[v], K = L.seg_start[v + 1] - s0;
    const uint32_t r0 = L.row_start[v], r1 = L.row_start[v + 1];
    const uint32_t p0 = r0 + ((uint32_t)u - s0) * (uint32_t)L.seg_len, p1 = min(p0 + (uint32_t)L.seg_len, r1);
    const float sum = lg_splat_segment(L, p0, p1, lane, CP, in);
    if (K == 1u) val[(size_t)v * CP + lane] = sum; else part[u * CP + lane] = sum;
}
// One group of LPV lanes (the power of two >= CP, at most a wave) per vertex: with 21 labels padded to 24 a whole wave
// per vertex would idle 40 of its 64 lanes.  No cross-lane traffic, so the groups of a wave are independent.
// vertex groups [0, Mb] belong to the bilateral lattice (row Mb = its zero row), the following Mg+1 to the Gaussian one
__global__ __launch_bounds__(256) void lg_splat2_kernel(LargeLattice Lb, LargeLattice Lg, int CP, int lpv_shift,
                                                         const float *__restrict__ in_b, const float *__restrict__ in_g,
                                                         float *__restrict__ val_b, float *__restrict__ val_g,
                                                         float *__restrict__ part_b, float *__restrict__ part_g) {
    const int lane = threadIdx.x & ((1 << lpv_shift) - 1);
    if (lane >= CP) return;
    const int Tb = Lb.M[1], Tg = Lg.M[1];
    const unsigned upb = blockDim.x >> lpv_shift;                 // units per workgroup
    // per XCD: first its strip of the bilateral lattice's segments, then its strip of the Gaussian lattice's (the same image strip)
    const unsigned nb_b = xcd_strip_blocks((size_t)Tb + 1, upb) / 8;
    const unsigned x = blockIdx.x & 7u, sl = blockIdx.x >> 3;
    size_t end;
    if (sl < nb_b) {
        const long long f = xcd_strip_first(x | (sl << 3), (size_t)Tb + 1, upb, &end);
        const size_t u = (size_t)f + (threadIdx.x >> lpv_shift);
        if (f < 0 || u >= end) return;
        lg_splat_unit(Lb, u, lane, CP, in_b, val_b, part_b);
    } else {
        const long long f = xcd_strip_first(x | ((sl - nb_b) << 3), (size_t)Tg + 1, upb, &end);
        const size_t u = (size_t)f + (threadIdx.x >> lpv_shift);
        if (f < 0 || u >= end) return;
        lg_splat_unit(Lg, u, lane, CP, in_g, val_g, part_g);
    }
}
// the vertices whose row was cut into several segments: value = ((p_0 + p_1) + p_2) + ...
__global__ __launch_bounds__(256) void lg_combine_kernel(LargeLattice Lb, LargeLattice Lg, int CP, int lpv_shift,
                                                          float *__restrict__ val_b, float *__restrict__ val_g,
                                                          const float *__restrict__ part_b, const float *__restrict__ part_g) {
    const int lane = threadIdx.x & ((1 << lpv_shift) - 1);
    if (lane >= CP) return;
    size_t i = (size_t)blockIdx.x * (blockDim.x >> lpv_shift) + (threadIdx.x >> lpv_shift);
    const int nb = Lb.M[2], ng = Lg.M[2];
    const LargeLattice *L = &Lb;
    float *val = val_b;
    const float *part = part_b;
    if (i >= (size_t)nb) { i -= nb; if (i >= (size_t)ng) return; L = &Lg; val = val_g; part = part_g; }
    const uint32_t v = L->multi_v[i];
    const uint32_t s0 = L->seg_start[v], s1 = L->seg_start[v + 1];
    float acc = part[(size_t)s0 * CP + lane];
    for (uint32_t q = s0 + 1; q < s1; q++) acc = acc + part[(size_t)q * CP + lane];
    val[(size_t)v * CP + lane] = acc;
}
// seq: <= 2 label planes take Permutohedral::seqCompute's arithmetic (permutohedral.cpp:476-527 via :600-601): the blur is
// summed in double (the literal 0.5) and the slice multiplies (w * value) * alpha
__device__ __forceinline__ float lg_blur_seq(float x0, float x1, float x2) {
    const float s = x1 + x2;
    return (float)((double)x0 + 0.5 * (double)s);
}
__device__ __forceinline__ void lg_blur_elem(const LargeLattice &L, int M, size_t v, int q, int CP4, int j, int seq,
                                             const float4 *__restrict__ a, float4 *__restrict__ b) {
    if (v == (size_t)M) { b[(size_t)M * CP4 + q] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
    const uint32_t n1 = L.nb1[(size_t)j * L.Mcap + v], n2 = L.nb2[(size_t)j * L.Mcap + v];
    const float4 x0 = a[v * CP4 + q], x1 = a[(size_t)n1 * CP4 + q], x2 = a[(size_t)n2 * CP4 + q];
    float4 o;
    if (seq) {
        o.x = lg_blur_seq(x0.x, x1.x, x2.x); o.y = lg_blur_seq(x0.y, x1.y, x2.y);
        o.z = lg_blur_seq(x0.z, x1.z, x2.z); o.w = lg_blur_seq(x0.w, x1.w, x2.w);
        b[v * CP4 + q] = o;
        return;
    }
    { float s = x1.x + x2.x; s = 0.5f * s; o.x = x0.x + s; }
    { float s = x1.y + x2.y; s = 0.5f * s; o.y = x0.y + s; }
    { float s = x1.z + x2.z; s = 0.5f * s; o.z = x0.z + s; }
    { float s = x1.w + x2.w; s = 0.5f * s; o.w = x0.w + s; }
    b[v * CP4 + q] = o;
}
#if defined(DSRG_EXP) && (DSRG_EXP & 1)       // experiment build: the linear work map of rounds 3-5 (make EXP=1 EXPSRC=lattice_large exp)
#define DSRG_BLUR_STRIPS 0
#endif
#ifndef DSRG_BLUR_STRIPS
#define DSRG_BLUR_STRIPS 1
#endif
constexpr bool kBlurStrips = DSRG_BLUR_STRIPS != 0;
// axis j of the bilateral lattice and, while j < 3, of the Gaussian lattice
__global__ void lg_blur2_kernel(LargeLattice Lb, LargeLattice Lg, int CP4, int j, int seq, const float4 *__restrict__ ab,
                                float4 *__restrict__ bb, const float4 *__restrict__ ag, float4 *__restrict__ bg) {
    const int Mb = *Lb.M, Mg = *Lg.M;
    if (!kBlurStrips) {
        const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        size_t v = idx / CP4;
        const int q = (int)(idx - v * CP4);
        if (v <= (size_t)Mb) { lg_blur_elem(Lb, Mb, v, q, CP4, j, seq, ab, bb); return; }
        v -= (size_t)Mb + 1;
        if (j < 3 && v <= (size_t)Mg) lg_blur_elem(Lg, Mg, v, q, CP4, j, seq, ag, bg);
        return;
    }
    // per XCD: its strip of the bilateral lattice's rows, then its strip of the Gaussian lattice's (vertex ids are image order: a
    // vertex and its two neighbours along an axis lie in one image strip, so the three rows meet in one L2 — as the splat's and the
    // slice's work maps, above)
    const unsigned upb = blockDim.x / (unsigned)CP4 * (unsigned)CP4;      // threads of a block that hold (row, quad) units: whole rows only
    const unsigned rows_pb = upb / (unsigned)CP4;
    if (threadIdx.x >= upb) return;
    const unsigned nb_b = xcd_strip_blocks((size_t)Mb + 1, rows_pb) / 8;
    const unsigned x = blockIdx.x & 7u, sl = blockIdx.x >> 3;
    const unsigned rl = threadIdx.x / (unsigned)CP4;
    const int q = (int)(threadIdx.x - rl * (unsigned)CP4);
    size_t end;
    if (sl < nb_b) {
        const long long f = xcd_strip_first(x | (sl << 3), (size_t)Mb + 1, rows_pb, &end);
        const size_t v = (size_t)f + rl;
        if (f >= 0 && v < end) lg_blur_elem(Lb, Mb, v, q, CP4, j, seq, ab, bb);
    } else if (j < 3) {
        const long long f = xcd_strip_first(x | ((sl - nb_b) << 3), (size_t)Mg + 1, rows_pb, &end);
        const size_t v = (size_t)f + rl;
        if (f >= 0 && v < end) lg_blur_elem(Lg, Mg, v, q, CP4, j, seq, ag, bg);
    }
}
__device__ __forceinline__ float4 lg_slice_elem(const LargeLattice &L, int D1, size_t i, int q, int CP4, int seq,
                                                const float4 *__restrict__ val, float neg_w) {
    const float alpha = 1.0f / (1.0f + exp2f(-(float)(D1 - 1)));
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < D1; r++) {
        const float4 x = val[(size_t)L.vid[(size_t)r * L.N + i] * CP4 + q];
        if (seq) {
            const float w0 = L.bary[(size_t)r * L.N + i];
            { float t = w0 * x.x; t = t * alpha; acc.x = acc.x + t; }
            { float t = w0 * x.y; t = t * alpha; acc.y = acc.y + t; }
            { float t = w0 * x.z; t = t * alpha; acc.z = acc.z + t; }
            { float t = w0 * x.w; t = t * alpha; acc.w = acc.w + t; }
            continue;
        }
        const float w = L.bary[(size_t)r * L.N + i] * alpha;
        acc.x = acc.x + w * x.x; acc.y = acc.y + w * x.y; acc.z = acc.z + w * x.z; acc.w = acc.w + w * x.w;
    }
    const float nv = L.norm[i];
    float4 o;
    o.x = neg_w * (acc.x * nv); o.y = neg_w * (acc.y * nv); o.z = neg_w * (acc.z * nv); o.w = neg_w * (acc.w * nv);
    return o;
}
// (Nimg, Npimg): batch mode — row i of the padded side is pixel i % Npimg of image i / Npimg, the caller's side holds the
// images' Nimg pixels back to back; the slots' padding pixels read as zero rows and are never written out.  Npimg = 0: one image.
__device__ __forceinline__ long long lg_user_row(size_t i, int Nimg, int Npimg) {
    if (Npimg == 0) return (long long)i;
    const size_t b = i / (size_t)Npimg, il = i - b * (size_t)Npimg;
    return il < (size_t)Nimg ? (long long)(b * (size_t)Nimg + il) : -1;
}
// slice both lattices, subtract the messages from -unary (Gaussian first) and renormalise.  kSlicePix pixels per 256-thread
// workgroup (403 workgroups of 256 pixels left the chip at 6 waves per CU: latency-bound gathers); the fp64-rounded exps run
// over all threads (pixel x label), the column maximum and the label-order sum per pixel.
constexpr int kSlicePix = 64;
__global__ __launch_bounds__(256) void lg_slice_update_kernel(LargeLattice Lb, LargeLattice Lg, int C, int CP,
                                                              const float4 *__restrict__ val_b, const float4 *__restrict__ val_g,
                                                              float neg_wb, float neg_wg, const float *__restrict__ neg_unary,
                                                              float *__restrict__ q_out, float *__restrict__ qn_b,
                                                              float *__restrict__ qn_g, int32_t *__restrict__ lab, int Nimg, int Npimg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *row = reinterpret_cast<float *>(smem);            // [kSlicePix][CP+1]
    const int P = CP + 1, CP4 = CP / 4, N = Lb.N;
    float *mxs = row + kSlicePix * P;                        // [kSlicePix] column maxima
    size_t strip_end;
    const long long first = xcd_strip_first(blockIdx.x, (size_t)N, kSlicePix, &strip_end);     // pixel strips per XCD
    if (first < 0) return;
    const size_t i0 = (size_t)first;
    const int npix = (int)min((size_t)kSlicePix, strip_end - i0);
    for (int k = threadIdx.x; k < npix * CP4; k += 256) {
        const int pl = k / CP4, q4 = k - pl * CP4;
        const size_t i = i0 + pl;
        const float4 tg = lg_slice_elem(Lg, 3, i, q4, CP4, C <= 2, val_g, neg_wg);
        const float4 tb = lg_slice_elem(Lb, 6, i, q4, CP4, C <= 2, val_b, neg_wb);
        const float4 nu = reinterpret_cast<const float4 *>(neg_unary)[i * CP4 + q4];
        float *r = row + pl * P + q4 * 4;
        { float v = nu.x; v = v - tg.x; v = v - tb.x; r[0] = v; }
        { float v = nu.y; v = v - tg.y; v = v - tb.y; r[1] = v; }
        { float v = nu.z; v = v - tg.z; v = v - tb.z; r[2] = v; }
        { float v = nu.w; v = v - tg.w; v = v - tb.w; r[3] = v; }
    }
    __syncthreads();
    if ((int)threadIdx.x < npix) {                            // expAndNormalize (densecrf.cpp:98-106): column maximum
        const float *r = row + threadIdx.x * P;
        float mx = -INFINITY;
        for (int c = 0; c < C; c++) mx = fmaxf(mx, r[c]);
        mxs[threadIdx.x] = mx;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < npix * C; k += 256) {      // exp of every (pixel, label)
        const int pl = k / C, c = k - pl * C;
        float *r = row + pl * P;
        r[c] = exp_cr(r[c] - mxs[pl]);
    }
    __syncthreads();
    if ((int)threadIdx.x < npix) {                            // label-order sum, division
        float *r = row + threadIdx.x * P;
        float sum = 0.0f;
        for (int c = 0; c < C; c++) sum = sum + r[c];
        for (int c = 0; c < C; c++) r[c] = r[c] / sum;
        for (int c = C; c < CP; c++) r[c] = 0.0f;
        if (lab) {                                           // last iteration: the MAP label too (lg_argmax_rows_kernel's rule: first maximum),
            const long long u = lg_user_row(i0 + threadIdx.x, Nimg, Npimg);      // so that reading the map costs no pass over Q
            if (u >= 0) {
                int m = 0;
                float best = r[0];
                for (int c = 1; c < C; c++) if (r[c] > best) { best = r[c]; m = c; }
                lab[u] = m;
            }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < npix * CP; k += 256) {
        const float v = row[(k / CP) * P + (k % CP)];
        if (q_out) q_out[i0 * CP + k] = v;                   // (only the last iteration's Q is read by anybody: 96 bytes a pixel less before it,
        if (!kSplatNormOnTheFly && qn_b) {                   //  and nobody splats after the last: 192 bytes less there)
            qn_b[i0 * CP + k] = v * Lb.norm[i0 + k / CP];  // in = Q * norm (pairwise.cpp:66) for the next splat
            qn_g[i0 * CP + k] = v * Lg.norm[i0 + k / CP];
        }
    }
}

__global__ void lg_pad_rows_kernel(int N, int C, int CP, const float *__restrict__ in, float *__restrict__ out, int negate,
                                   int Nimg, int Npimg) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * CP) return;
    const size_t i = idx / CP;
    const int c = (int)(idx - i * CP);
    const long long u = lg_user_row(i, Nimg, Npimg);
    const float v = (c < C && u >= 0) ? in[(size_t)u * C + c] : 0.0f;
    out[idx] = negate ? -v : v;
}
__global__ void lg_unpad_rows_kernel(int N, int C, int CP, const float *__restrict__ in, float *__restrict__ out, int Nimg, int Npimg) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * C) return;
    const size_t i = idx / C;
    const int c = (int)(idx - i * C);
    const long long u = lg_user_row(i, Nimg, Npimg);
    if (u >= 0) out[(size_t)u * C + c] = in[i * CP + c];
}
__global__ void lg_argmax_rows_kernel(int N, int C, int CP, const float *__restrict__ q, int32_t *__restrict__ lab, int Nimg, int Npimg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const long long u = lg_user_row((size_t)i, Nimg, Npimg);
    if (u < 0) return;
    int m = 0;
    float best = q[(size_t)i * CP];
    for (int c = 1; c < C; c++) { const float v = q[(size_t)i * CP + c]; if (v > best) { best = v; m = c; } }
    lab[u] = m;
}

// ---------------------------------------------------------------------------------------------
// The two device-wide primitives of the build, written for its sizes (the full-resolution lattices have 0.1 - 1.5 M entries):
//
// Exclusive sum of n uint32: blocks of 8192 elements (1024 threads x 8); kernel 1 leaves every block's total, kernel 2 lets
// every block add up the totals in front of it (at most a few hundred: one load per thread + a block reduction) and scan its
// own elements (serial over a thread's eight, wave shuffles, wave totals through LDS).
constexpr int kScanT = 1024, kScanV = 8, kScanBlock = kScanT * kScanV;

__device__ __forceinline__ uint32_t lg_block_reduce(uint32_t v, uint32_t *red /* [16] */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_down((int)v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < kScanT / 64; w++) t += red[w];
    __syncthreads();
    return t;
}
__global__ __launch_bounds__(kScanT) void lg_scan_totals_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ totals, int n) {
    __shared__ uint32_t red[16];
    const int base = blockIdx.x * kScanBlock + threadIdx.x * kScanV;
    uint32_t s = 0;
#pragma unroll
    for (int u = 0; u < kScanV; u++) s += base + u < n ? in[base + u] : 0u;
    s = lg_block_reduce(s, red);
    if (threadIdx.x == 0) totals[blockIdx.x] = s;
}
__global__ __launch_bounds__(kScanT) void lg_scan_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                         const uint32_t *__restrict__ totals, int n) {
    __shared__ uint32_t red[16], wave_sum[16];
    uint32_t before = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += kScanT) before += totals[b];
    before = lg_block_reduce(before, red);
    const int base = blockIdx.x * kScanBlock + threadIdx.x * kScanV;
    uint32_t v[kScanV], s = 0;
#pragma unroll
    for (int u = 0; u < kScanV; u++) { v[u] = base + u < n ? in[base + u] : 0u; s += v[u]; }
    uint32_t incl = s;                                       // inclusive scan of the threads' sums inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
        if ((int)(threadIdx.x & 63) >= o) incl += t;
    }
    if ((threadIdx.x & 63) == 63) wave_sum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t off = before + incl - s;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) off += wave_sum[w];
#pragma unroll
    for (int u = 0; u < kScanV; u++) {
        if (base + u < n) out[base + u] = off;
        off += v[u];
    }
}
static int lg_scan_blocks(int n) { return (n + kScanBlock - 1) / kScanBlock; }
// tmp: lg_scan_blocks(n) uint32
static int lg_exclusive_sum(const uint32_t *in, uint32_t *out, int n, uint32_t *tmp, hipStream_t s) {
    const int nb = lg_scan_blocks(n);
    hipLaunchKernelGGL(lg_scan_totals_kernel, dim3(nb), dim3(kScanT), 0, s, in, tmp, n);
    hipLaunchKernelGGL(lg_scan_kernel, dim3(nb), dim3(kScanT), 0, s, in, out, tmp, n);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

// Stable sort of (key, value) pairs by the low `bits` bits of the key (the vertex ids: 14 - 18 significant bits), least
// significant digit first, 8 bits per pass.  Per pass: digit histograms of blocks of 2048 pairs -> exclusive sum over
// (digit-major, block-minor) -> scatter.  Stability inside a block: its 32 chunks of 64 pairs are ranked chunk by chunk
// (a lane's rank among the lanes of its chunk with the same digit from eight ballots; the chunks' digit counts scanned in
// chunk order by one thread per digit).
constexpr int kSortT = 256, kSortChunks = 32, kSortBlock = 64 * kSortChunks;
__device__ __forceinline__ unsigned long long lg_same_digit_lanes(uint32_t d, bool valid) {
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const unsigned long long has = __ballot((d >> b) & 1u);
        m &= ((d >> b) & 1u) ? has : ~has;
    }
    return m;
}
__global__ __launch_bounds__(kSortT) void lg_sort_hist_kernel(const uint32_t *__restrict__ keys, uint32_t *__restrict__ hist, int n, int shift,
                                                              int nblk) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * kSortBlock;
    for (int i = threadIdx.x; i < kSortBlock; i += kSortT)
        if (base + i < n) atomicAdd(&h[(keys[base + i] >> shift) & 255u], 1u);
    __syncthreads();
    hist[(size_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}
__global__ __launch_bounds__(kSortT) void lg_sort_scatter_kernel(const uint32_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                                 uint32_t *__restrict__ kout, uint32_t *__restrict__ vout,
                                                                 const uint32_t *__restrict__ offs, int n, int shift, int nblk) {
    __shared__ uint16_t cnt[kSortChunks][256];              // pairs of chunk c with digit d, then their first rank in the block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < kSortChunks * 256; i += kSortT) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const int base = blockIdx.x * kSortBlock;
    uint32_t key[kSortChunks / 4], val[kSortChunks / 4], rank[kSortChunks / 4];
#pragma unroll
    for (int q = 0; q < kSortChunks / 4; q++) {            // wave w takes chunks w, w + 4, ...
        const int c = wave + 4 * q, i = base + c * 64 + lane;
        const bool valid = i < n;
        key[q] = valid ? kin[i] : 0u;
        val[q] = valid ? vin[i] : 0u;
        const uint32_t d = (key[q] >> shift) & 255u;
        const unsigned long long same = lg_same_digit_lanes(d, valid);
        rank[q] = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        if (valid && rank[q] == 0) cnt[c][d] = (uint16_t)__popcll(same);
    }
    __syncthreads();
    {   // one thread per digit: exclusive scan of its counts over the chunks (block-relative; the lanes below add where the
        // block's pairs of the digit start globally)
        uint32_t rel = 0;
        for (int c = 0; c < kSortChunks; c++) {
            const uint32_t t = cnt[c][threadIdx.x];
            cnt[c][threadIdx.x] = (uint16_t)rel;
            rel += t;
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kSortChunks / 4; q++) {
        const int c = wave + 4 * q, i = base + c * 64 + lane;
        if (i < n) {
            const uint32_t d = (key[q] >> shift) & 255u;
            const uint32_t pos = offs[(size_t)d * nblk + blockIdx.x] + cnt[c][d] + rank[q];
            kout[pos] = key[q];
            vout[pos] = val[q];
        }
    }
}
static int lg_sort_blocks(int n) { return (n + kSortBlock - 1) / kSortBlock; }
static size_t lg_sort_tmp_words(int n) {                   // histogram + its scan + the scan's block totals
    const size_t h = (size_t)256 * lg_sort_blocks(n);
    return 2 * h + (size_t)lg_scan_blocks((int)h) + 64;
}
// sorts by key bits [0, bits); the result is in (kout, vout); (kin, vin) are used as the other ping-pong buffer
static int lg_sort_pairs(uint32_t *tmp, uint32_t *kin, uint32_t *kout, uint32_t *vin, uint32_t *vout, int n, int bits, hipStream_t s) {
    const int nblk = lg_sort_blocks(n), passes = (bits + 7) / 8;
    uint32_t *hist = tmp, *offs = tmp + (size_t)256 * nblk, *tot = offs + (size_t)256 * nblk;
    uint32_t *ka = kin, *va = vin, *kb = kout, *vb = vout;
    // an even number of passes ends in (kin, vin): start from the other pair then, so that the result lands in (kout, vout)
    for (int p = 0; p < passes; p++) {
        hipLaunchKernelGGL(lg_sort_hist_kernel, dim3(nblk), dim3(kSortT), 0, s, ka, hist, n, 8 * p, nblk);
        if (int rc = lg_exclusive_sum(hist, offs, 256 * nblk, tot, s)) return rc;
        hipLaunchKernelGGL(lg_sort_scatter_kernel, dim3(nblk), dim3(kSortT), 0, s, ka, va, kb, vb, offs, n, 8 * p, nblk);
        uint32_t *t = ka; ka = kb; kb = t;
        t = va; va = vb; vb = t;
    }
    DSRG_LAUNCH_CHECK();
    if (ka != kout) {                                       // even number of passes: the result sits in the input pair
        DSRG_HIP_CHECK(hipMemcpyAsync(kout, ka, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, s));
        DSRG_HIP_CHECK(hipMemcpyAsync(vout, va, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, s));
    }
    return DSRG_OK;
}

// ---------------------------------------------------------------------------------------------
struct LargeCrf {
    int W, H, C, CP, N;                        // N: pixel slots of the object (W * H, or nimg * Npimg in batch mode)
    int nimg, Nimg, Npimg;                     // images per call (1 = the plain object), pixels per image, per slot (0 when nimg == 1)
    LargeLattice Lg, Lb;
    void *arena;
    uint32_t *prim_tmp; size_t prim_words;     // scratch of the scan / sort primitives above
    hipStream_t stream;                        // every launch and copy of this object (dsrg_crf_set_stream; default: the null stream)
    bool async;                                // the entry points do not wait for the stream (the caller does: dsrg_crf_synchronize)
    float *neg_unary, *q;                  // [N][CP]
    float *qn_b, *qn_g;                    // [N][CP] q * norm of the bilateral / Gaussian kernel: the splat inputs
    bool lab_valid;                        // `lab` holds the arg-max of the current Q (written by the last slice / update launch)
    float *val_a, *val_b;                  // ping-pong [(Mb+1) + (Mg+1)][CP], grown on demand: bilateral rows first
    size_t val_rows;
    float *part;                           // per-segment partial sums of the splat [T_b + T_g][CP], grown on demand
    size_t part_rows;
    float *val1_a, *val1_b;                // one-channel buffers for the norm pass [Mcap+1]
    unsigned char *im;                     // [N][3]
    int32_t *lab;
    float *stage;                          // [N][C] host-layout staging
    bool lattices_valid;                   // Lg/Lb were built for the current image and kernel widths
    dsrg_crf_params built_for;
    Profiler prof;                         // optional HIP-event brackets around the splat launches (dsrg_crf_profile_*)
};

static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
static inline int blocks_for(size_t n, int t) { return (int)((n + t - 1) / t); }

static size_t large_lattice_carve(LargeLattice &L, unsigned char *p, int d, int N) {
    const int d1 = d + 1, KW = (d * 16 + 31) / 32;
    L.d = d; L.N = N; L.Npad = (N + 3) / 4 * 4; L.E = N * d1; L.Epad = L.Npad * d1; L.Mcap = L.Epad;
    int cap = 1024;
    while (cap < 2 * L.Epad) cap <<= 1;
    L.cap = cap; L.M_host = 0;
    size_t off = 0;
    auto take = [&](size_t bytes) { unsigned char *r = p ? p + off : nullptr; off += al(bytes); return r; };
    L.key_e = (uint32_t *)take(sizeof(uint32_t) * (size_t)L.Epad * KW);
    L.table = (uint32_t *)take(sizeof(uint32_t) * (size_t)cap);
    L.slot_e = (uint32_t *)take(sizeof(uint32_t) * (size_t)L.Epad);
    // (+2: the same arrays later hold seg_start / seg_cnt, Mcap + 1 = Epad + 1 entries)
    L.first = (uint32_t *)take(sizeof(uint32_t) * ((size_t)L.Epad + 2));
    L.scanned = (uint32_t *)take(sizeof(uint32_t) * ((size_t)L.Epad + 2));
    L.key_v = (uint32_t *)take(sizeof(uint32_t) * (size_t)L.Mcap * KW);
    L.vid = (uint32_t *)take(sizeof(uint32_t) * (size_t)L.E);
    L.nb1 = (uint32_t *)take(sizeof(uint32_t) * (size_t)d1 * L.Mcap);
    L.nb2 = (uint32_t *)take(sizeof(uint32_t) * (size_t)d1 * L.Mcap);
    L.ent_vid = (uint32_t *)take(sizeof(uint32_t) * (size_t)L.E);
    L.ent_idx = (uint32_t *)take(sizeof(uint32_t) * (size_t)L.E);
    L.srt_vid = (uint32_t *)take(sizeof(uint32_t) * (size_t)L.E);
    L.srt_idx = (uint32_t *)take(sizeof(uint32_t) * (size_t)L.E);
    L.cnt = (uint32_t *)take(sizeof(uint32_t) * (size_t)(L.Mcap + 1));
    L.row_start = (uint32_t *)take(sizeof(uint32_t) * (size_t)(L.Mcap + 1));
    L.csr_pix = (uint32_t *)take(sizeof(uint32_t) * (size_t)L.E);
    L.bary = (float *)take(sizeof(float) * (size_t)L.E);
    L.csr_w = (float *)take(sizeof(float) * (size_t)L.E);
    L.norm = (float *)take(sizeof(float) * (size_t)N);
    L.M = (int *)take(sizeof(int) * 4);
    L.seg_start = L.first; L.seg_cnt = L.scanned; L.seg_v = L.slot_e; L.multi_v = L.key_e;        // dead once lg_vid_kernel has run
    L.T_host = L.nmulti_host = 0;
    L.seg_len = [] { const char *e = getenv("DSRG_SPLAT_SEG"); const int v = e ? atoi(e) : 0; return v >= 8 && v <= 65536 ? v : kSplatSeg; }();   // (tools: A/B)
    L.nimg = 1; L.Nimg = N; L.Npimg = 0;
    return off;
}

int large_crf_create(int W, int H, int C, LargeCrf **out, int nimages) {
    if (nimages < 1 || nimages > kLargeBatchMax)
        return set_error(DSRG_ERR_UNSUPPORTED, "1 .. %d images per batched CRF object", kLargeBatchMax);
    const long long slot = nimages > 1 ? ((long long)W * H + 3) / 4 * 4 : (long long)W * H;
    if (slot * nimages * 6 >= (1ll << 31) / 4) return set_error(DSRG_ERR_UNSUPPORTED, "map too large");
    LargeCrf *c = new (std::nothrow) LargeCrf();
    if (!c) return set_error(DSRG_ERR_NOMEM, "host allocation failed");
    memset(c, 0, sizeof(*c));
    c->W = W; c->H = H; c->C = C; c->CP = (C + 3) & ~3;
    c->nimg = nimages; c->Nimg = W * H; c->Npimg = nimages > 1 ? (int)slot : 0;
    c->N = (int)(slot * nimages);
    const int N = c->N;
    LargeLattice tmp;
    const size_t sg = large_lattice_carve(tmp, nullptr, 2, N), sb = large_lattice_carve(tmp, nullptr, 5, N);
    const int Emax = N * 6, Mcap5 = ((N + 3) / 4 * 4) * 6;
    c->prim_words = lg_sort_tmp_words(Emax) + (size_t)lg_scan_blocks(Mcap5 + 2) + 64;
    const size_t rows = sizeof(float) * (size_t)N * c->CP;
    const size_t total = sg + sb + al(sizeof(uint32_t) * c->prim_words) + 4 * al(rows) + 2 * al(sizeof(float) * (size_t)(Mcap5 + 1)) +
                         al((size_t)N * 3) + al(sizeof(int32_t) * (size_t)N) + al(sizeof(float) * (size_t)N * C);
    hipError_t e = hipMalloc(&c->arena, total);
    if (e != hipSuccess) { delete c; return set_error(DSRG_ERR_NOMEM, "hipMalloc(%zu) failed: %s", total, hipGetErrorString(e)); }
    unsigned char *p = (unsigned char *)c->arena;
    p += large_lattice_carve(c->Lg, p, 2, N);
    p += large_lattice_carve(c->Lb, p, 5, N);
    c->Lg.nimg = c->Lb.nimg = c->nimg; c->Lg.Nimg = c->Lb.Nimg = c->Nimg; c->Lg.Npimg = c->Lb.Npimg = c->Npimg;
    c->prim_tmp = (uint32_t *)p; p += al(sizeof(uint32_t) * c->prim_words);
    c->neg_unary = (float *)p; p += al(rows);
    c->q = (float *)p; p += al(rows);
    c->qn_b = (float *)p; p += al(rows);
    c->qn_g = (float *)p; p += al(rows);
    c->val1_a = (float *)p; p += al(sizeof(float) * (size_t)(Mcap5 + 1));
    c->val1_b = (float *)p; p += al(sizeof(float) * (size_t)(Mcap5 + 1));
    c->im = p; p += al((size_t)N * 3);
    c->lab = (int32_t *)p; p += al(sizeof(int32_t) * (size_t)N);
    c->stage = (float *)p; p += al(sizeof(float) * (size_t)N * C);
    *out = c;
    return DSRG_OK;
}

void large_crf_destroy(LargeCrf *c) {
    if (!c) return;
    for (int i = 0; i < c->prof.cap; i++) { (void)hipEventDestroy(c->prof.start[i]); (void)hipEventDestroy(c->prof.stop[i]); }
    delete[] c->prof.start; delete[] c->prof.stop;
    if (c->arena) (void)hipFree(c->arena);
    if (c->val_a) (void)hipFree(c->val_a);
    if (c->val_b) (void)hipFree(c->val_b);
    if (c->part) (void)hipFree(c->part);
    delete c;
}

template <int D>
static int large_build(LargeCrf *c, LargeLattice &L, const LatticeFeat &F, hipStream_t s) {
    constexpr int D1 = D + 1;
    const int T = 256;
    DSRG_HIP_CHECK(hipMemsetAsync(L.table, 0xFF, sizeof(uint32_t) * (size_t)L.cap, s));
    DSRG_HIP_CHECK(hipMemsetAsync(L.cnt, 0, sizeof(uint32_t) * (size_t)(L.Mcap + 1), s));
    DSRG_HIP_CHECK(hipMemsetAsync(L.M + 3, 0, sizeof(int), s));               // batch mode: "a key left the tagged range"
    hipLaunchKernelGGL(lg_embed_kernel<D>, dim3(blocks_for(L.Npad, T)), dim3(T), 0, s, L, F, c->im);
    hipLaunchKernelGGL(lg_insert_kernel<D>, dim3(blocks_for(L.Epad, T)), dim3(T), 0, s, L);
    hipLaunchKernelGGL(lg_first_kernel, dim3(blocks_for(L.Epad, T)), dim3(T), 0, s, L);
    if (int rc = lg_exclusive_sum(L.first, L.scanned, L.Epad, c->prim_tmp, s)) return rc;
    hipLaunchKernelGGL(lg_assign_kernel<D>, dim3(blocks_for(L.Epad, T)), dim3(T), 0, s, L);
    hipLaunchKernelGGL(lg_vid_kernel, dim3(blocks_for(L.E, T)), dim3(T), 0, s, L, D1);
    DSRG_LAUNCH_CHECK();
    // CSR row starts and the splat segments, on worst-case grids (M is still on the device): the arrays they alias (first,
    // scanned, slot_e, key_e) are dead from here on
    if (int rc = lg_exclusive_sum(L.cnt, L.row_start, L.Mcap + 1, c->prim_tmp, s)) return rc;
    DSRG_HIP_CHECK(hipMemsetAsync(L.M + 1, 0, sizeof(int) * 2, s));
    hipLaunchKernelGGL(lg_seg_count_kernel, dim3(blocks_for((size_t)L.Mcap + 1, T)), dim3(T), 0, s, L);
    if (int rc = lg_exclusive_sum(L.seg_cnt, L.seg_start, L.Mcap + 1, c->prim_tmp, s)) return rc;
    hipLaunchKernelGGL(lg_seg_fill_kernel, dim3(blocks_for((size_t)L.Mcap + 1, T)), dim3(T), 0, s, L);
    DSRG_LAUNCH_CHECK();
    int mtn[4] = {0, 0, 0, 0};
    DSRG_HIP_CHECK(hipMemcpyAsync(mtn, L.M, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
    DSRG_HIP_CHECK(hipStreamSynchronize(s));                      // M and the segment count size the remaining launches
    L.M_host = mtn[0]; L.T_host = mtn[1]; L.nmulti_host = mtn[2];
    if (mtn[3] != 0)
        return set_error(DSRG_ERR_UNSUPPORTED, "batched CRF: a lattice coordinate of the d = %d kernel lies beyond +-2048 — the image "
                         "number does not fit its key; filter these images one at a time", D);
    // the aliasing above holds segments in slot_e (Epad words) and multi-segment vertices in key_e (Epad * KW words)
    if (L.M_host < 0 || L.M_host > L.Mcap || L.T_host < 0 || L.T_host > L.Epad || L.nmulti_host < 0 ||
        (long long)L.nmulti_host > (long long)L.Epad * KeyWords<D>::value)
        return set_error(DSRG_ERR_HIP, "lattice build out of range: M %d (cap %d), segments %d (cap %d), multi-segment vertices %d",
                         L.M_host, L.Mcap, L.T_host, L.Epad, L.nmulti_host);
    const int M = L.M_host;
    hipLaunchKernelGGL(lg_neigh_kernel<D>, dim3(blocks_for(M, T)), dim3(T), 0, s, L);
    int bits = 1;
    while ((1ll << bits) < (long long)M + 1) bits++;
    if (int rc = lg_sort_pairs(c->prim_tmp, L.ent_vid, L.srt_vid, L.ent_idx, L.srt_idx, L.E, bits, s)) return rc;
    hipLaunchKernelGGL(lg_csr_kernel, dim3(blocks_for(L.E, T)), dim3(T), 0, s, L, D1);
    // norm = 1/sqrt(K 1 + 1e-20)
    hipLaunchKernelGGL(lg_splat1_kernel, dim3(blocks_for(M, T)), dim3(T), 0, s, L, c->val1_a);
    float *a = c->val1_a, *b = c->val1_b;
    for (int j = 0; j < D1; j++) {
        hipLaunchKernelGGL(lg_blur1_kernel, dim3(blocks_for(M, T)), dim3(T), 0, s, L, j, a, b);
        float *t = a; a = b; b = t;
    }
    hipLaunchKernelGGL(lg_norm_kernel, dim3(blocks_for(L.N, T)), dim3(T), 0, s, L, D1, a);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

// `unary` / `im` / outputs may be host or device pointers (hipMemcpyDefault resolves the kind): the test-time pipeline
// keeps its scores on the GPU, the Cython-style callers pass numpy memory.
int large_crf_set_unary(LargeCrf *c, const float *unary) {
    DSRG_HIP_CHECK(hipMemcpyAsync(c->stage, unary, sizeof(float) * (size_t)c->nimg * c->Nimg * c->C, hipMemcpyDefault, c->stream));
    hipLaunchKernelGGL(lg_pad_rows_kernel, dim3(blocks_for((size_t)c->N * c->CP, 256)), dim3(256), 0, c->stream, c->N, c->C,
                       c->CP, c->stage, c->neg_unary, 1, c->Nimg, c->Npimg);
    DSRG_LAUNCH_CHECK();
    if (!c->async) DSRG_HIP_CHECK(hipStreamSynchronize(c->stream));         // the caller may reuse its (host) buffer
    return DSRG_OK;
}
int large_crf_zero_unary(LargeCrf *c) {
    DSRG_HIP_CHECK(hipMemsetAsync(c->neg_unary, 0, sizeof(float) * (size_t)c->N * c->CP, c->stream));
    return DSRG_OK;
}
int large_crf_set_image(LargeCrf *c, const unsigned char *im) {
    DSRG_HIP_CHECK(hipMemcpyAsync(c->im, im, (size_t)c->nimg * c->Nimg * 3, hipMemcpyDefault, c->stream));
    if (!c->async) DSRG_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->lattices_valid = false;
    return DSRG_OK;
}

// DenseCRF::inference on the large path; q ends up in c->q ([N][CP])
int large_crf_infer(LargeCrf *c, const dsrg_crf_params *prm, int n_iters) {
    hipStream_t s = c->stream;
    LatticeFeat Fg, Fb;
    lattice_feat_init(Fg, 2, c->W, c->H, prm->theta_gamma_x, prm->theta_gamma_y, 1.f, 1.f, 1.f);
    lattice_feat_init(Fb, 5, c->W, c->H, prm->theta_alpha_x, prm->theta_alpha_y, prm->theta_beta_r, prm->theta_beta_g,
                      prm->theta_beta_b);
    int rc = DSRG_OK;
    // the lattices belong to addPairwiseEnergy (densecrf.cpp:61-81): repeated inference() calls reuse them
    const bool same_kernels = c->lattices_valid && memcmp(&c->built_for, prm, offsetof(dsrg_crf_params, n_iters)) == 0;
    if (!same_kernels) {
        c->lattices_valid = false;
        rc = large_build<2>(c, c->Lg, Fg, s);
        if (rc) return rc;
        rc = large_build<5>(c, c->Lb, Fb, s);
        if (rc) return rc;
        c->built_for = *prm;
        c->lattices_valid = true;
    }
    const int Mb = c->Lb.M_host, Mg = c->Lg.M_host, CP = c->CP, CP4 = CP / 4;
    const size_t need = (size_t)Mb + 1 + (size_t)Mg + 1;
    if (need > c->val_rows) {
        if (c->val_a) (void)hipFree(c->val_a);
        if (c->val_b) (void)hipFree(c->val_b);
        c->val_a = c->val_b = nullptr;
        c->val_rows = 0;
        hipError_t e = hipMalloc(&c->val_a, sizeof(float) * need * CP);
        if (e == hipSuccess) e = hipMalloc(&c->val_b, sizeof(float) * need * CP);
        if (e != hipSuccess) return set_error(DSRG_ERR_NOMEM, "hipMalloc of lattice values failed: %s", hipGetErrorString(e));
        c->val_rows = need;
    }
    const int Tb = c->Lb.T_host, Tg = c->Lg.T_host, nmulti = c->Lb.nmulti_host + c->Lg.nmulti_host;
    const size_t need_part = (size_t)Tb + (size_t)Tg + 1;
    if (need_part > c->part_rows) {
        if (c->part) (void)hipFree(c->part);
        c->part = nullptr; c->part_rows = 0;
        hipError_t e = hipMalloc(&c->part, sizeof(float) * need_part * CP);
        if (e != hipSuccess) return set_error(DSRG_ERR_NOMEM, "hipMalloc of splat partials failed: %s", hipGetErrorString(e));
        c->part_rows = need_part;
    }
    const int T = 256;
    const size_t upd_lds = sizeof(float) * 256 * (size_t)(CP + 1);
    hipLaunchKernelGGL(lg_update_kernel, dim3(blocks_for(c->N, T)), dim3(T), upd_lds, s, c->N, c->C, CP, c->neg_unary,
                       nullptr, nullptr, 0, c->q, c->Lb.norm, c->Lg.norm, c->qn_b, c->qn_g);
    int lpv_shift = 3;
    while ((1 << lpv_shift) < CP && lpv_shift < 6) lpv_shift++;
    const size_t g_off = ((size_t)Mb + 1) * CP;                          // Gaussian rows follow the bilateral ones
    for (int it = 0; it < n_iters; it++) {
        const bool timed = c->prof.active && c->prof.used < c->prof.cap;      // the dominant kernel of this path (dsrg_crf_profile_*)
        if (timed) DSRG_HIP_CHECK(hipEventRecord(c->prof.start[c->prof.used], s));
        hipLaunchKernelGGL(lg_splat2_kernel,
                           dim3(xcd_strip_blocks((size_t)Tb + 1, 256 >> lpv_shift) + xcd_strip_blocks((size_t)Tg + 1, 256 >> lpv_shift)),
                           dim3(256), 0, s, c->Lb, c->Lg, CP, lpv_shift, kSplatNormOnTheFly ? c->q : c->qn_b,
                           kSplatNormOnTheFly ? c->q : c->qn_g, c->val_a, c->val_a + g_off, c->part,
                           c->part + (size_t)Tb * CP);
        if (nmulti > 0)
            hipLaunchKernelGGL(lg_combine_kernel, dim3(blocks_for((size_t)nmulti, 256 >> lpv_shift)), dim3(256), 0, s, c->Lb, c->Lg, CP,
                               lpv_shift, c->val_a, c->val_a + g_off, c->part, c->part + (size_t)Tb * CP);
        if (timed) { DSRG_HIP_CHECK(hipEventRecord(c->prof.stop[c->prof.used], s)); c->prof.used++; }
        float *a = c->val_a, *b = c->val_b;
        for (int j = 0; j < 6; j++) {
            const size_t rows = j < 3 ? need : (size_t)Mb + 1;
            const unsigned rows_pb = 256u / (unsigned)CP4;
            const unsigned nblk = kBlurStrips ? xcd_strip_blocks((size_t)Mb + 1, rows_pb) + (j < 3 ? xcd_strip_blocks((size_t)Mg + 1, rows_pb) : 0u)
                                              : (unsigned)blocks_for(rows * CP4, 256);
            hipLaunchKernelGGL(lg_blur2_kernel, dim3(nblk), dim3(256), 0, s, c->Lb, c->Lg, CP4, j,
                               c->C <= 2 ? 1 : 0, (const float4 *)a, (float4 *)b, (const float4 *)(a + g_off), (float4 *)(b + g_off));
            float *t = a; a = b; b = t;
        }
        // after 6 swaps the bilateral result is back in val_a; the Gaussian one stopped after 3 swaps, in val_b
        hipLaunchKernelGGL(lg_slice_update_kernel, dim3(xcd_strip_blocks((size_t)c->N, kSlicePix)), dim3(T),
                           sizeof(float) * (kSlicePix * (size_t)(CP + 1) + kSlicePix), s, c->Lb, c->Lg, c->C, CP,
                           (const float4 *)c->val_a, (const float4 *)(c->val_b + g_off), -prm->w_bilateral,
                           -prm->w_gaussian, c->neg_unary, (it == n_iters - 1 || kSplatNormOnTheFly) ? c->q : nullptr,
                           it == n_iters - 1 ? nullptr : c->qn_b, it == n_iters - 1 ? nullptr : c->qn_g,
                           it == n_iters - 1 ? c->lab : nullptr, c->Nimg, c->Npimg);
    }
    c->lab_valid = n_iters > 0;
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

int large_crf_read_q(LargeCrf *c, float *out_host) {
    hipLaunchKernelGGL(lg_unpad_rows_kernel, dim3(blocks_for((size_t)c->N * c->C, 256)), dim3(256), 0, c->stream, c->N, c->C,
                       c->CP, c->q, c->stage, c->Nimg, c->Npimg);
    DSRG_LAUNCH_CHECK();
    DSRG_HIP_CHECK(hipMemcpyAsync(out_host, c->stage, sizeof(float) * (size_t)c->nimg * c->Nimg * c->C, hipMemcpyDefault, c->stream));
    if (!c->async) DSRG_HIP_CHECK(hipStreamSynchronize(c->stream));    // the result is the caller's when the call returns
    return DSRG_OK;
}
int large_crf_read_map(LargeCrf *c, int32_t *labels_host) {
    if (!c->lab_valid)                                       // (no iteration ran: Q is the softmax of the unaries, lg_update_kernel)
        hipLaunchKernelGGL(lg_argmax_rows_kernel, dim3(blocks_for(c->N, 256)), dim3(256), 0, c->stream, c->N, c->C, c->CP, c->q, c->lab,
                       c->Nimg, c->Npimg);
    DSRG_LAUNCH_CHECK();
    DSRG_HIP_CHECK(hipMemcpyAsync(labels_host, c->lab, sizeof(int32_t) * (size_t)c->nimg * c->Nimg, hipMemcpyDefault, c->stream));
    if (!c->async) DSRG_HIP_CHECK(hipStreamSynchronize(c->stream));
    return DSRG_OK;
}
void large_crf_set_stream(LargeCrf *c, hipStream_t s, bool async) { c->stream = s; c->async = async; }
int large_crf_lattice_size(LargeCrf *c, int k) { return k == 0 ? c->Lg.M_host : c->Lb.M_host; }
int large_crf_images(LargeCrf *c) { return c->nimg; }
Profiler *large_crf_profiler(LargeCrf *c) { return &c->prof; }

}  // namespace dsrg
