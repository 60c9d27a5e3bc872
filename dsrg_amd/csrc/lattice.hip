// Permutohedral lattice construction on gfx950: one 1024-thread workgroup per
// (image, kernel) lattice, hash table resident in LDS.
//
// Replaces Permutohedral::init + HashTable (CRF/src/permutohedral.cpp:54-131,
// 140-321, the SSE variant an x86 build of the reference takes) and
// DenseKernel::initLattice (CRF/src/pairwise.cpp:40-62).  The embedding follows
// the reference's fp32 operation order exactly (this file is compiled with
// -ffp-contract=off); vertex ids differ from the reference's insertion order,
// which no result depends on: the splat CSR built here lists every vertex's
// contributions in the reference's accumulation order (pixel-major).
#include <math.h>
#include "common.h"
#include "embed.h"

namespace dsrg {

// hash table: 16-bit slots (entry index, later vertex id; 0xFFFF = empty), capacity >= 2x the
// worst-case vertex count so linear-probing chains stay short
__host__ __device__ inline int lattice_table_cap(int Mcap) {
    int cap = 2048;
    while (cap < 2 * Mcap && cap < 65536) cap <<= 1;
    return cap;
}

// compact, exactly comparable form of a vertex key for the LDS-resident key array:
//   D = 2: the packed word itself;  D = 5: five 12-bit fields (|coordinate| < 2048, checked)
template <int D> struct CompactKey;
template <> struct CompactKey<2> {
    using type = uint32_t;
    __device__ static __forceinline__ uint32_t make(const uint32_t (&w)[1]) { return w[0]; }
};
template <> struct CompactKey<5> {
    using type = unsigned long long;
    __device__ static __forceinline__ unsigned long long make(const uint32_t (&w)[3]) {
        const unsigned long long c0 = (w[0] & 0xFFFFu) - 0x7800u, c1 = (w[0] >> 16) - 0x7800u;
        const unsigned long long c2 = (w[1] & 0xFFFFu) - 0x7800u, c3 = (w[1] >> 16) - 0x7800u;
        const unsigned long long c4 = (w[2] & 0xFFFFu) - 0x7800u;
        return (c0 & 0xFFFu) | ((c1 & 0xFFFu) << 12) | ((c2 & 0xFFFu) << 24) | ((c3 & 0xFFFu) << 36) | ((c4 & 0xFFFu) << 48);
    }
};

// the packed key (embed.h: 16-bit fields biased by 0x8000) back from its compact form — valid whenever make() was (12-bit range)
__device__ __forceinline__ void unpack_compact(uint32_t (&w)[1], uint32_t c) { w[0] = c; }
__device__ __forceinline__ void unpack_compact(uint32_t (&w)[3], unsigned long long c) {
    const uint32_t f0 = (uint32_t)(c & 0xFFFu), f1 = (uint32_t)((c >> 12) & 0xFFFu), f2 = (uint32_t)((c >> 24) & 0xFFFu);
    const uint32_t f3 = (uint32_t)((c >> 36) & 0xFFFu), f4 = (uint32_t)((c >> 48) & 0xFFFu);
    w[0] = (f0 + 0x7800u) | ((f1 + 0x7800u) << 16);
    w[1] = (f2 + 0x7800u) | ((f3 + 0x7800u) << 16);
    w[2] = f4 + 0x7800u;
}

constexpr int kBuildVPT = 32;   // largest instantiation: Mcap <= 32*1024 (vertices / entries per thread)

void *g_build_dbg = nullptr;     // tools only: 16 u64 phase timestamps per lattice (dsrg_debug_set_build_trace)

// ---------------------------------------------------------------------------------
// Stage 0 of the build: embed every pixel (incl. the SSE zero padding) -> simplex keys + barycentric weights
// (Permutohedral::init, permutohedral.cpp:185-259), one pixel per thread over the whole chip.  Inside the single-workgroup
// build this was 11 of 60 us: ~800 instructions per pixel on one CU.  With float images the colour is the on-the-fly
// resampling of CRFLayer (pylayers.py:70-75), which saves that kernel too.
constexpr int kEmbedWG = 256;
template <int D>
__global__ __launch_bounds__(kEmbedWG) void lattice_embed_kernel(LatticeView L, LatticeFeat F, LatticeColours col) {
    constexpr int D1 = D + 1, KW = KeyWords<D>::value;
    using ckey_t = typename CompactKey<D>::type;
    const int b = blockIdx.y, i = blockIdx.x * kEmbedWG + threadIdx.x;
    const int N = L.N, Npad = (N + 3) & ~3, Epad = Npad * D1;
    int bad = 0;
    if (i < Npad) {
        uint32_t c = 0;
        if (D == 5 && i < N) {
            if (col.images) {
                c = map_pixel_rgb(col.images, b, col.Hi, col.Wi, F.H, F.W, i);
                unsigned char *o = col.im_out + ((size_t)b * N + i) * 3;
                o[0] = (unsigned char)c; o[1] = (unsigned char)(c >> 8); o[2] = (unsigned char)(c >> 16);
            } else {
                const unsigned char *px = col.im_u8 + ((size_t)b * N + i) * 3;
                c = (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16);
            }
        }
        uint32_t keys[D1][KW];
        float bc[D1];
        bad = embed_pixel_rgb<D>(F, i, N, (float)(c & 0xFFu), (float)((c >> 8) & 0xFFu), (float)((c >> 16) & 0xFFu), keys, bc);
        uint32_t *key_e = L.key_e + ((size_t)b * Epad + (size_t)i * D1) * KW;
        ckey_t *ck = reinterpret_cast<ckey_t *>(L.ckeys_e) + (size_t)b * Epad + (size_t)i * D1;
        float *bary = L.bary + (size_t)b * D1 * N;
#pragma unroll
        for (int r = 0; r <= D; r++) {
#pragma unroll
            for (int q = 0; q < KW; q++) key_e[r * KW + q] = keys[r][q];
            if (i < N) bary[(size_t)r * N + i] = bc[r];
            ck[r] = CompactKey<D>::make(keys[r]);
        }
    }
    const int any = __syncthreads_or(bad & 1) | (__syncthreads_or(bad & 2) ? 2 : 0);
    if (threadIdx.x == 0) L.embed_bad[(size_t)b * 32 + blockIdx.x] = any;
}

// ---------------------------------------------------------------------------------
// CSR of the splat, contributions in reference order (entry index ascending), and from it the filter kernel's first-term /
// extras lists.  One 1024-thread workgroup per lattice; `ev` = the vertex id of the thread's entries tid + k*1024.  Runs
// inside lattice_build_kernel when the build is one kernel, and as EXTRA workgroups of the neighbour-search launch when it
// is split: neither stage needs the other's result, and the single-workgroup build was 44 us with it.  There it is shared by
// `nparts` workgroups per lattice: each counts and scans all rows (cheap) and fills, sorts and writes out the rows of its
// range of vertices [M part / nparts, M (part + 1) / nparts).
template <int D, int EPT>
__device__ __forceinline__ void lattice_csr_phase(const LatticeView &L, int b, int M, unsigned char *smem,
                                                  const uint16_t (&ev)[EPT], int wl_in_lds, int split, int part, int nparts,
                                                  unsigned long long *dbg) {
#define DSRG_STAMP(i_) do { if (dbg && threadIdx.x == 0) dbg[(size_t)b * 16 + (i_)] = wall_clock64(); } while (0)
    constexpr int D1 = D + 1;
    const int tid = threadIdx.x;
    const int N = L.N, E = N * D1, Mcap = L.Mcap;
    const uint16_t *vid = L.vid + (size_t)b * D1 * N;
    const float *bary = L.bary + (size_t)b * D1 * N;
    uint16_t *row_start = L.row_start + (size_t)b * (Mcap + 2);
    float *csr_w = L.csr_w + (size_t)b * E;
    uint16_t *first_pix = L.first_pix + (size_t)b * Mcap, *x_pix = L.x_pix + (size_t)b * E;
    float *first_w = L.first_w + (size_t)b * Mcap, *x_w = L.x_w + (size_t)b * E;
    uint32_t *cnt = reinterpret_cast<uint32_t *>(smem);                       // [Mcap+1]
    uint16_t *csr_e = reinterpret_cast<uint16_t *>(smem + (((size_t)(Mcap + 1) * 4 + 15) & ~(size_t)15));   // [E]
    int *scan2 = reinterpret_cast<int *>(reinterpret_cast<unsigned char *>(csr_e) + (((size_t)E * 2 + 15) & ~(size_t)15));
    for (int v = tid; v <= Mcap; v += kWG) cnt[v] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; k++)
        if (tid + k * kWG < E) atomicAdd(&cnt[ev[k]], 1u);
    __syncthreads();
    {
        const int per = (Mcap + kWG - 1) / kWG;
        const int v0 = tid * per, v1 = min(v0 + per, M);
        int local = 0;
        for (int v = v0; v < v1; v++) local += (int)cnt[v];
        int tot;
        int run = block_exclusive_scan(local, scan2, &tot);
        for (int v = v0; v < v1; v++) {
            int c = (int)cnt[v];
            cnt[v] = (uint32_t)run;
            if (part == 0) row_start[v] = (uint16_t)run;
            run += c;
        }
        if (tid == 0 && part == 0) row_start[M] = (uint16_t)E;
    }
    __syncthreads();
    if (part == 0) DSRG_STAMP(5);
    const int vlo = (int)((long long)M * part / nparts), vhi = (int)((long long)M * (part + 1) / nparts);
    const int p0 = vlo < M ? (int)cnt[vlo] : E;       // start of my first row (cnt[v] = START of row v until the fill below)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; k++) {
        const int e = tid + k * kWG;
        if (e < E && (int)ev[k] >= vlo && (int)ev[k] < vhi) csr_e[atomicAdd(&cnt[ev[k]], 1u)] = (uint16_t)e;
    }
    __syncthreads();
    const int p1 = vhi > vlo ? (int)cnt[vhi - 1] : p0;   // end of my last row
    for (int v = vlo + tid; v < vhi; v += kWG) {     // cnt[v] is now the END of segment v (for the rows of my range)
        const int s = v == vlo ? p0 : (int)cnt[v - 1], t = (int)cnt[v];
        for (int a = s + 1; a < t; a++) {         // insertion sort, segments are short
            uint16_t x = csr_e[a];
            int c = a - 1;
            while (c >= s && csr_e[c] > x) { csr_e[c + 1] = csr_e[c]; c--; }
            csr_e[c + 1] = x;
        }
    }
    __syncthreads();
    if (part == 0) DSRG_STAMP(6);
    // weights of the sorted entries: to HBM for the filter kernel, and (when it fits) to LDS for the
    // norm pass below
    float *wl = wl_in_lds ? reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(scan2) + 32 * 4) : csr_w;
    // The filter kernel's view of the same lists: the FIRST contributor of every vertex, vertex-indexed (most rows have
    // exactly one entry), and the remaining entries as one compact list — rows without entries are the SSE padding's
    // phantom vertices, which were created last and therefore sit at the end of the id range, so the extras of row v start
    // at row_start[v] - v.
    {
        // (the weight and the vertex of every sorted entry come back from global memory: all of a thread's loads in flight
        // before the first use — one round trip instead of one per entry)
        float pw[EPT];
        uint16_t pv[EPT], pi[EPT];
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            const int pos = p0 + tid + k * kWG;
            const int e = pos < p1 ? csr_e[pos] : 0;
            const int i = e / D1, r = e - i * D1;
            pi[k] = (uint16_t)i;
            pw[k] = bary[(size_t)r * N + i];
            pv[k] = vid[(size_t)r * N + i];
        }
        const bool keep_csr_w = !(split && D == 5);            // its only reader is the d = 2 / unsplit norm pass
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            const int pos = p0 + tid + k * kWG;
            if (pos < p1) {
                const float w = pw[k];
                const int v = pv[k], i = pi[k];
                if (keep_csr_w) csr_w[pos] = w;
                if (wl_in_lds) wl[pos] = w;
                const int start = v == vlo ? p0 : (int)cnt[v - 1];     // cnt[v] = END of row v
                if (pos == start) { first_pix[v] = (uint16_t)i; first_w[v] = w; }
                else { x_pix[pos - v - 1] = (uint16_t)i; x_w[pos - v - 1] = w; }
            }
        }
    }
    if (part == 0) DSRG_STAMP(11);
    {
        // rows without entries (the phantom vertices) form the tail of the id range: the first of them gives the number of
        // vertices with entries, hence the number of extras
        int *first_empty = scan2;                              // (the scan scratch is free here)
        if (tid == 0) *first_empty = M;
        __syncthreads();
        for (int v = vlo + tid; v < vhi; v += kWG) {
            const int start = v == vlo ? p0 : (int)cnt[v - 1];
            if ((int)cnt[v] == start) {                        // phantom vertex: contributes an exact 0
                first_pix[v] = 0; first_w[v] = 0.0f;
                atomicMin(first_empty, v);
            }
        }
        __syncthreads();
        // (several parts: the build kernel zeroed the word; the part that holds the first empty row supplies the maximum)
        if (tid == 0) { if (nparts == 1) L.nextra[b] = E - *first_empty; else atomicMax(&L.nextra[b], E - *first_empty); }
    }
    __syncthreads();
    if (part == 0) DSRG_STAMP(7);
#undef DSRG_STAMP
}

template <int D, int VPT>   // VPT >= ceil(Mcap / 1024): vertices (and entries) per thread
__global__ __launch_bounds__(kWG) void lattice_build_kernel(LatticeView L, int cap,
                                                              int wl_in_lds, int lds_keys, int split, unsigned long long *dbg) {
#define DSRG_STAMP(i_) do { if (dbg && threadIdx.x == 0) dbg[(size_t)blockIdx.x * 16 + (i_)] = wall_clock64(); } while (0)
    DSRG_STAMP(0);
    constexpr int D1 = D + 1, KW = KeyWords<D>::value;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int N = L.N, Npad = (N + 3) & ~3, E = N * D1, Epad = Npad * D1, Mcap = L.Mcap;
    const uint32_t mask = (uint32_t)cap - 1u;

    using ckey_t = typename CompactKey<D>::type;
    uint16_t *tab = reinterpret_cast<uint16_t *>(smem);                   // [cap] 16-bit slots
    uint32_t *tabw = reinterpret_cast<uint32_t *>(smem);                  // the same, as words (CAS granularity)
    int *scan_scratch = reinterpret_cast<int *>(smem + (size_t)cap * 2);  // [32]
    uint32_t *bm = reinterpret_cast<uint32_t *>(smem + (size_t)cap * 2 + 32 * 4);   // [Epad/32 + 1] first-occurrence bitmap
    uint32_t *wp = bm + (Epad / 32 + 1);                                  // [Epad/32 + 1] word prefix
    ckey_t *ckeys = reinterpret_cast<ckey_t *>(smem + (((size_t)cap * 2 + 32 * 4 + 2 * (size_t)(Epad / 32 + 1) * 4 + 15) & ~(size_t)15));   // [Mcap]

    uint16_t *vid = L.vid + (size_t)b * D1 * N;
    float *bary = L.bary + (size_t)b * D1 * N;
    uint32_t *nb = L.nb + (size_t)b * D1 * Mcap;
    float *csr_w = L.csr_w + (size_t)b * E;
    float *norm = L.norm + (size_t)b * N;
    uint32_t *key_e = L.key_e + (size_t)b * Epad * KW;
    uint32_t *key_v = L.key_v + (size_t)b * Mcap * KW;

    for (int h = tid; h < cap / 2; h += kWG) tabw[h] = 0xFFFFFFFFu;
    // ---- phase 1: the pixels were embedded by lattice_embed_kernel; its range flags, and the entries' compact keys into LDS
    // (the region becomes the per-VERTEX key array in phase 3): the probes of phase 2 then never leave the CU
    int key_range_bad = 0, key12_bad = 0;
    if (tid < (Npad + kEmbedWG - 1) / kEmbedWG) {
        const int f = L.embed_bad[(size_t)b * 32 + tid];
        key_range_bad = f & 1; key12_bad = (f >> 1) & 1;
    }
    if (lds_keys) {
        const ckey_t *ke = reinterpret_cast<const ckey_t *>(L.ckeys_e) + (size_t)b * Epad;
        if ((reinterpret_cast<uintptr_t>(ke) & 15) == 0 && ((size_t)Epad * sizeof(ckey_t)) % 16 == 0)
            stage16_to_lds<5>(reinterpret_cast<unsigned char *>(ckeys), ke, (uint32_t)((size_t)Epad * sizeof(ckey_t)), tid);
        else
            for (int q = tid; q < Epad; q += kWG) ckeys[q] = ke[q];
    }
    const int compact_ok = (D == 2) ? 1 : !__syncthreads_or(key12_bad);    // (also the phase barrier)
    if (D == 2) __syncthreads();
    const bool fast_keys = lds_keys && compact_ok;
    DSRG_STAMP(1);

    // ---- phase 2: deduplicate keys (open addressing, linear probing, table in LDS).  A slot ends up
    // holding the SMALLEST entry index of its key = the key's first occurrence in the reference's
    // visiting order (pixel-major, corner-minor; permutohedral.cpp:261-276).  Slots are 16 bits wide
    // (two per CAS word).  Thread t owns entries t + k*1024.
    constexpr uint32_t kEmpty = 0xFFFFu;
    constexpr int EPT = VPT;                              // entries per thread (Epad <= Mcap)
    uint16_t hs[EPT];                                     // slot of my k-th entry
    // (entry after entry.  Measured and dropped: probing five entries of a thread together — table words, keys and CASes
    // each in flight for all five — 21.4 instead of 12.6 us for this phase (every round costs the wave the work of all its
    // lanes' pending entries, and failed CASes multiply); every lane advancing through its own entries at its own pace, one
    // probe step per loop trip — 15.2 us (the per-trip bookkeeping outweighs the shorter critical path))
    if (fast_keys) {
        ckey_t mine_k[EPT];
#pragma unroll
        for (int k = 0; k < EPT; k++) mine_k[k] = ckeys[min(tid + k * kWG, Epad - 1)];      // all reads in flight together
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            const int e = tid + k * kWG;
            hs[k] = 0;
            if (e < Epad) {
                uint32_t w[KW];
                unpack_compact(w, mine_k[k]);
                uint32_t h = hash_key<KW>(w) & mask;                 // the same hash the neighbour search computes
                for (;;) {
                    const uint32_t sh = (h & 1u) * 16u;
                    const uint32_t cur = *reinterpret_cast<volatile uint32_t *>(&tabw[h >> 1]);
                    const uint32_t half = (cur >> sh) & 0xFFFFu;
                    const uint32_t mine = (cur & ~(0xFFFFu << sh)) | ((uint32_t)e << sh);
                    if (half == kEmpty) {
                        if (atomicCAS(&tabw[h >> 1], cur, mine) == cur) break;
                        continue;                          // the word changed under us: look again
                    }
                    if (ckeys[half] == mine_k[k]) {
                        if ((uint32_t)e < half && atomicCAS(&tabw[h >> 1], cur, mine) != cur) continue;
                        break;
                    }
                    h = (h + 1) & mask;
                }
                hs[k] = (uint16_t)h;
            }
        }
    } else {
#pragma unroll
    for (int k = 0; k < EPT; k++) {
        const int e = tid + k * kWG;
        hs[k] = 0;
        if (e < Epad) {
            uint32_t w[KW];
            load_key<KW>(w, key_e + (size_t)e * KW);
            uint32_t h = hash_key<KW>(w) & mask;
            for (;;) {
                const uint32_t sh = (h & 1u) * 16u;
                const uint32_t cur = *reinterpret_cast<volatile uint32_t *>(&tabw[h >> 1]);
                const uint32_t half = (cur >> sh) & 0xFFFFu;
                const uint32_t mine = (cur & ~(0xFFFFu << sh)) | ((uint32_t)e << sh);
                if (half == kEmpty) {
                    if (atomicCAS(&tabw[h >> 1], cur, mine) == cur) break;
                    continue;                              // the word changed under us: look again
                }
                if (key_eq<KW>(w, key_e + (size_t)half * KW)) {
                    if ((uint32_t)e < half && atomicCAS(&tabw[h >> 1], cur, mine) != cur) continue;
                    break;
                }
                h = (h + 1) & mask;
            }
            hs[k] = (uint16_t)h;
        }
    }
    }
    __syncthreads();
    DSRG_STAMP(2);

    // ---- phase 3: vertex ids in first-occurrence order — exactly the ids the reference's hash table
    // hands out (HashTable::find(create), permutohedral.cpp:98-111).  A bitmap of "first occurrence"
    // flags over the entries is prefix-summed in LDS.
    int M;
    {
        const int nwords = Epad / 32 + 1;
        if (tid == 0 && nwords - 1 >= EPT * (kWG / 32)) bm[nwords - 1] = 0;      // the one word no ballot below covers
        uint32_t isfirst = 0;
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            // entries tid + k*1024: a wave's 64 lanes are 64 consecutive entries = two bitmap words, written whole
            const int e = tid + k * kWG;
            const bool f = e < Epad && (uint32_t)tab[hs[k]] == (uint32_t)e;
            isfirst |= (f ? 1u : 0u) << k;
            const unsigned long long m = __ballot(f);
            if ((tid & 31) == 0 && (e >> 5) < nwords) bm[e >> 5] = (tid & 32) ? (uint32_t)(m >> 32) : (uint32_t)m;
        }
        // the keys of my first-occurrence entries leave the entry-indexed array before it is overwritten vertex-indexed
        ckey_t first_key[EPT];
        if (fast_keys) {
#pragma unroll
            for (int k = 0; k < EPT; k++) first_key[k] = ckeys[min(tid + k * kWG, Epad - 1)];
        }
        __syncthreads();
        // exclusive prefix of the per-word popcounts (nwords <= 1024 for Epad <= 32768)
        const int myc = tid < nwords ? __popc(bm[tid]) : 0;
        const int pre = block_exclusive_scan(myc, scan_scratch, &M);
        if (tid < nwords) wp[tid] = (uint32_t)pre;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            const int e = tid + k * kWG;
            if ((isfirst >> k) & 1u) {
                const uint32_t id = wp[e >> 5] + (uint32_t)__popc(bm[e >> 5] & ((1u << (e & 31)) - 1u));
                uint32_t w[KW];
                if (fast_keys) unpack_compact(w, first_key[k]);
                else load_key<KW>(w, key_e + (size_t)e * KW);
#pragma unroll
                for (int t = 0; t < KW; t++) key_v[(size_t)id * KW + t] = w[t];
                if (fast_keys) ckeys[id] = first_key[k];
                tab[hs[k]] = (uint16_t)id;
            }
        }
        if (tid == 0) L.M[b] = M;
    }
    __syncthreads();
    DSRG_STAMP(3);

    // ---- phase 4: vertex id of every real (pixel, corner) entry
    uint16_t ev[EPT];                                     // ... kept for the CSR passes below
#pragma unroll
    for (int k = 0; k < EPT; k++) {
        const int e = tid + k * kWG;
        ev[k] = 0;
        if (e < E) {
            const int i = e / D1, r = e - i * D1;
            ev[k] = tab[hs[k]];
            vid[(size_t)r * N + i] = ev[k];
        }
    }

    if (split) {
        // the neighbour search and the normalisation run in follow-up kernels spread over more workgroups
        // (lattice_neigh_kernel / lattice_norm_kernel): hand them the hash table and the compact keys
        __syncthreads();
        uint32_t *tg = L.tab_g + (size_t)b * (cap / 2);
        for (int q = tid; q < cap / 2; q += kWG) tg[q] = tabw[q];
        if (fast_keys) {
            ckey_t *kg = reinterpret_cast<ckey_t *>(L.ckeys_g) + (size_t)b * Mcap;
            for (int q = tid; q < M; q += kWG) kg[q] = ckeys[q];
        }
        const int any_bad = __syncthreads_or(key_range_bad);
        // bit 0 ("diagonal": no vertex shared, no blur neighbour) is set on trust here and cleared by the neighbour search
        if (tid == 0) { L.flags[b] = (any_bad ? 2 : 0) | (fast_keys ? 8 : 0) | (M == E ? 1 : 0); L.nextra[b] = 0; }
    } else {
    // ---- phase 5: blur neighbours (permutohedral.cpp:303-318): 2(d+1) hash look-ups per vertex,
        // advanced together one probe per round.  With the compact keys resident in LDS a probe is two
        // LDS reads (slot, key) and no global traffic; otherwise the candidates of a round are confirmed
        // by one batch of key fetches from HBM/L2.
        int has_nb = 0;
#pragma unroll 1
        for (int k = 0; k < VPT; k++) {
            const int v = tid + k * kWG;
            if (v >= M) break;
            uint32_t w[KW];
            load_key<KW>(w, key_v + (size_t)v * KW);
            uint32_t word[D1];
#pragma unroll
            for (int j = 0; j < D1; j++) word[j] = 0;
            // two halves of D1 look-ups each (keeps the probe state in registers):
            //   half 0: n1 = all coordinates -1, axis j +d;   half 1: n2 = all +1, axis j -d
#pragma unroll 1
            for (int half = 0; half < 2; half++) {
                uint32_t hq[D1], found[D1];
                ckey_t qc[D1];
#pragma unroll
                for (int j = 0; j < D1; j++) {
                    uint32_t q[KW];
                    neighbour_key<D>(q, w, j, half != 0);
                    hq[j] = hash_key<KW>(q) & mask;
                    qc[j] = CompactKey<D>::make(q);
                    found[j] = (uint32_t)M;                      // "no neighbour" = the zero sentinel slot M
                }
                uint32_t pend = (1u << D1) - 1u;
                if (fast_keys) {
                    while (pend) {
                        uint32_t t[D1];
#pragma unroll
                        for (int j = 0; j < D1; j++) t[j] = tab[hq[j]];
                        ckey_t ck[D1];
#pragma unroll
                        for (int j = 0; j < D1; j++) ck[j] = ckeys[min(t[j], (uint32_t)Mcap - 1u)];
#pragma unroll
                        for (int j = 0; j < D1; j++) {
                            if ((pend >> j) & 1u) {
                                if (t[j] == kEmpty) pend &= ~(1u << j);
                                else if (ck[j] == qc[j]) { found[j] = t[j]; pend &= ~(1u << j); }
                                else hq[j] = (hq[j] + 1) & mask;
                            }
                        }
                    }
                } else {
                    while (pend) {
                        uint32_t t[D1];
#pragma unroll
                        for (int j = 0; j < D1; j++) t[j] = tab[hq[j]];
                        uint32_t kv[D1][KW];
#pragma unroll
                        for (int j = 0; j < D1; j++)
                            load_key<KW>(kv[j], key_v + (size_t)min(t[j], (uint32_t)Mcap - 1u) * KW);
#pragma unroll
                        for (int j = 0; j < D1; j++) {
                            if ((pend >> j) & 1u) {
                                uint32_t q[KW];
                                neighbour_key<D>(q, w, j, half != 0);
                                bool eq = true;
#pragma unroll
                                for (int x = 0; x < KW; x++) eq &= (kv[j][x] == q[x]);
                                if (t[j] == kEmpty) pend &= ~(1u << j);
                                else if (eq) { found[j] = t[j]; pend &= ~(1u << j); }
                                else hq[j] = (hq[j] + 1) & mask;
                            }
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < D1; j++) {
                    word[j] |= found[j] << (half * 16);
                    has_nb |= found[j] != (uint32_t)M;
                }
            }
#pragma unroll
            for (int j = 0; j < D1; j++) nb[(size_t)j * Mcap + v] = word[j];
        }
        // the unused tail of every axis points at the sentinel too, so that consumers need no v < M test on the words
        for (int v = M + tid; v < Mcap; v += kWG) {
#pragma unroll
            for (int j = 0; j < D1; j++) nb[(size_t)j * Mcap + v] = (uint32_t)M | ((uint32_t)M << 16);
        }
        {   // diagonal lattice: every entry owns its vertex and no vertex has a blur neighbour
            const int any_nb = __syncthreads_or(has_nb);
            const int any_bad = __syncthreads_or(key_range_bad);
            // bit 0: diagonal; bit 1: a key coordinate left the range the packed neighbour arithmetic covers
            // bit 0: diagonal; bit 1: a key coordinate left the range the packed neighbour arithmetic covers;
            // bit 2: every vertex has exactly one contributor (M == E: entry e IS vertex e's only splat term)
            if (tid == 0) L.flags[b] = ((!any_nb && M == E) ? 1 : 0) | (any_bad ? 2 : 0);      // bit 2 is added in phase 7
        }
    }
    DSRG_STAMP(4);

    // ---- phase 6: CSR of the splat (lattice_csr_phase).  In a split build it runs beside the neighbour search instead.
    if (split) { DSRG_STAMP(10); return; }
    lattice_csr_phase<D, EPT>(L, b, M, smem, ev, wl_in_lds, split, 0, 1, dbg);
    uint32_t *cnt = reinterpret_cast<uint32_t *>(smem);                       // [Mcap+1] (now the END of every row)
    uint16_t *csr_e = reinterpret_cast<uint16_t *>(smem + (((size_t)(Mcap + 1) * 4 + 15) & ~(size_t)15));   // [E]
    int *scan2 = reinterpret_cast<int *>(reinterpret_cast<unsigned char *>(csr_e) + (((size_t)E * 2 + 15) & ~(size_t)15));
    float *wl = wl_in_lds ? reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(scan2) + 32 * 4) : csr_w;

    if (!split) {
    // ---- phase 7: norm = 1/sqrt(K 1 + 1e-20)  (pairwise.cpp:44,54-57), one channel through
        // Permutohedral::seqCompute (permutohedral.cpp:476-527): blur evaluated in double
        float *val = reinterpret_cast<float *>(smem);                             // [Mcap+1], aliases cnt
        int multi_rows = 0;
        {
            float s0[VPT];
#pragma unroll
            for (int k = 0; k < VPT; k++) {
                const int v = tid + k * kWG;
                float s = 0.0f;
                if (v < M) {
                    const uint32_t a0 = v == 0 ? 0u : cnt[v - 1], z0 = cnt[v];    // cnt[v] = END of row v
                    for (uint32_t pos = a0; pos < z0; pos++) s = s + wl[pos] * 1.0f;
                    multi_rows |= (z0 - a0 != 1u);
                }
                s0[k] = s;
            }
            __syncthreads();                                                      // cnt is dead from here
#pragma unroll
            for (int k = 0; k < VPT; k++) {
                const int v = tid + k * kWG;
                if (v < M) val[v] = s0[k];
            }
            if (tid == 0) val[M] = 0.0f;                                          // zero sentinel = "no neighbour"
        }
        __syncthreads();
        DSRG_STAMP(8);
        {
            const rsrc_t r_nb = make_rsrc(nb, sizeof(uint32_t) * (size_t)D1 * Mcap);
            for (int j = 0; j <= D; j++) {
                uint32_t word[VPT];
#pragma unroll
                for (int k = 0; k < VPT; k++)      // unconditional, all in flight (past-the-end reads 0)
                    word[k] = (k * kWG < Mcap) ? ld_u32(r_nb, (uint32_t)tid * 4u, (uint32_t)j * (uint32_t)Mcap * 4u + (uint32_t)k * (kWG * 4u)) : 0u;
                float nv[VPT];
#pragma unroll
                for (int k = 0; k < VPT; k++) {
                    const int v = tid + k * kWG;
                    const bool ok = v < M;
                    const int n1 = ok ? (int)(word[k] & 0xffffu) : M, n2 = ok ? (int)(word[k] >> 16) : M;
                    const float s = val[n1] + val[n2];
                    nv[k] = (float)((double)val[ok ? v : M] + 0.5 * (double)s);
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < VPT; k++) {
                    const int v = tid + k * kWG;
                    if (v < M) val[v] = nv[k];
                }
                __syncthreads();
            }
        }
        DSRG_STAMP(9);
        const float alpha = 1.0f / (1.0f + exp2f((float)-D));
        for (int i = tid; i < N; i += kWG) {
            float out = 0.0f;
#pragma unroll
            for (int r = 0; r <= D; r++) {
                float t = bary[(size_t)r * N + i] * val[vid[(size_t)r * N + i]];
                t = t * alpha;
                out = out + t;
            }
            norm[i] = (float)(1.0 / sqrt((double)out + 1e-20));
        }
        const int any_multi = __syncthreads_or(multi_rows);
        if (tid == 0 && M == E && !any_multi) L.flags[b] |= 4;                    // every vertex has exactly one contributor
    }
    DSRG_STAMP(10);
#undef DSRG_STAMP
}

// ---------------------------------------------------------------------------------
// Split build, stage 2: blur neighbours (permutohedral.cpp:303-318) with nsplit workgroups per
// lattice.  Each workgroup reloads the hash table (and the compact keys) into LDS and resolves the
// 2(d+1) look-ups of its share of the vertices; probes never leave LDS when the compact keys exist.
// nsplit >= ceil(Mcap / 1024): no thread gets a second vertex (41x41: with 8 workgroups a chunk of 1038 sent every workgroup
// round its loop twice for 14 threads' sake)

constexpr int kCsrParts = 2;      // CSR workgroups per lattice in the neighbour-search launch
template <int D, int VPT>
__global__ __launch_bounds__(kWG) void lattice_neigh_kernel(LatticeView L, int cap, int lds_keys, int nsplit,
                                                            unsigned long long *dbg) {
    constexpr int D1 = D + 1, KW = KeyWords<D>::value;
    using ckey_t = typename CompactKey<D>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x / (nsplit + kCsrParts), part = blockIdx.x % (nsplit + kCsrParts), tid = threadIdx.x;
    if (part >= nsplit) {
        // one of the lattice's CSR workgroups (see lattice_csr_phase): the entries' vertex ids come back from the build's vid array
        const int N = L.N, E = N * D1;
        const uint16_t *vid = L.vid + (size_t)b * D1 * N;
        uint16_t ev[VPT];
#pragma unroll
        for (int k = 0; k < VPT; k++) {
            const int e = tid + k * kWG;
            ev[k] = 0;
            if (e < E) { const int i = e / D1, r = e - i * D1; ev[k] = vid[(size_t)r * N + i]; }
        }
        lattice_csr_phase<D, VPT>(L, b, L.M[b], smem, ev, 0, 1, part - nsplit, kCsrParts, dbg);
        return;
    }
#define DSRG_STAMP(i_) do { if (dbg && threadIdx.x == 0) dbg[64 * 16 + ((size_t)b * nsplit + part) * 8 + (i_)] = wall_clock64(); } while (0)
    DSRG_STAMP(0);
    const int Mcap = L.Mcap, M = L.M[b];
    const uint32_t mask = (uint32_t)cap - 1u;
    constexpr uint32_t kEmpty = 0xFFFFu;
    uint16_t *tab = reinterpret_cast<uint16_t *>(smem);
    ckey_t *ckeys = reinterpret_cast<ckey_t *>(smem + (size_t)cap * 2);
    const bool fast_keys = lds_keys && (L.flags[b] & 8);
    // staging from a cold L2: every load of a round in flight before its first LDS store (element loops of 4 / 8 bytes per
    // thread and trip took 9 + 8 dependent round trips)
    stage16_to_lds<4>(smem, L.tab_g + (size_t)b * (cap / 2), (uint32_t)cap * 2u, tid);
    if (fast_keys) {
        const ckey_t *kg = reinterpret_cast<const ckey_t *>(L.ckeys_g) + (size_t)b * Mcap;
        if ((reinterpret_cast<uintptr_t>(kg) & 15) == 0)
            stage16_to_lds<5>(reinterpret_cast<unsigned char *>(ckeys), kg, ((uint32_t)M * (uint32_t)sizeof(ckey_t) + 15u) & ~15u, tid);
        else
            for (int q = tid; q < M; q += kWG) ckeys[q] = kg[q];
    }
    __syncthreads();
    DSRG_STAMP(1);
    int has_nb = 0;
    const uint32_t *key_v = L.key_v + (size_t)b * Mcap * KW;
    uint32_t *nb = L.nb + (size_t)b * D1 * Mcap;
    const int chunk = (M + nsplit - 1) / nsplit;
    const int v_end = min(M, (part + 1) * chunk);
    for (int v = part * chunk + tid; v < v_end; v += kWG) {
        uint32_t w[KW];
        load_key<KW>(w, key_v + (size_t)v * KW);
        uint32_t word[D1];
#pragma unroll
        for (int j = 0; j < D1; j++) word[j] = 0;
#pragma unroll 1
        for (int half = 0; half < 2; half++) {          // half 0: n1 (all -1, axis j +d); half 1: n2 (all +1, axis j -d)
            uint32_t hq[D1], found[D1];
            ckey_t qc[D1];
#pragma unroll
            for (int j = 0; j < D1; j++) {
                uint32_t q[KW];
                neighbour_key<D>(q, w, j, half != 0);
                hq[j] = hash_key<KW>(q) & mask;
                qc[j] = CompactKey<D>::make(q);
                found[j] = (uint32_t)M;                          // "no neighbour" = the zero sentinel slot M
            }
            uint32_t pend = (1u << D1) - 1u;
            if (fast_keys) {
                while (pend) {
                    uint32_t t[D1];
#pragma unroll
                    for (int j = 0; j < D1; j++) t[j] = tab[hq[j]];
                    ckey_t ck[D1];
#pragma unroll
                    for (int j = 0; j < D1; j++) ck[j] = ckeys[min(t[j], (uint32_t)Mcap - 1u)];
#pragma unroll
                    for (int j = 0; j < D1; j++) {
                        if ((pend >> j) & 1u) {
                            if (t[j] == kEmpty) pend &= ~(1u << j);
                            else if (ck[j] == qc[j]) { found[j] = t[j]; pend &= ~(1u << j); }
                            else hq[j] = (hq[j] + 1) & mask;
                        }
                    }
                }
            } else {
                while (pend) {
                    uint32_t t[D1];
#pragma unroll
                    for (int j = 0; j < D1; j++) t[j] = tab[hq[j]];
                    uint32_t kv[D1][KW];
#pragma unroll
                    for (int j = 0; j < D1; j++)
                        load_key<KW>(kv[j], key_v + (size_t)min(t[j], (uint32_t)Mcap - 1u) * KW);
#pragma unroll
                    for (int j = 0; j < D1; j++) {
                        if ((pend >> j) & 1u) {
                            uint32_t q[KW];
                            neighbour_key<D>(q, w, j, half != 0);
                            bool eq = true;
#pragma unroll
                            for (int x = 0; x < KW; x++) eq &= (kv[j][x] == q[x]);
                            if (t[j] == kEmpty) pend &= ~(1u << j);
                            else if (eq) { found[j] = t[j]; pend &= ~(1u << j); }
                            else hq[j] = (hq[j] + 1) & mask;
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < D1; j++) { word[j] |= found[j] << (half * 16); has_nb |= (found[j] != (uint32_t)M); }
            DSRG_STAMP(2 + half);
        }
#pragma unroll
        for (int j = 0; j < D1; j++) nb[(size_t)j * Mcap + v] = word[j];
    }
    // the unused tail of every axis points at the sentinel too (consumers need no v < M test on the words)
    const int tail = Mcap - M, tchunk = (tail + nsplit - 1) / nsplit;
    for (int v = M + part * tchunk + tid; v < min(Mcap, M + (part + 1) * tchunk); v += kWG) {
#pragma unroll
        for (int j = 0; j < D1; j++) nb[(size_t)j * Mcap + v] = (uint32_t)M | ((uint32_t)M << 16);
    }
    DSRG_STAMP(4);
    if (__syncthreads_or(has_nb) && tid == 0) atomicAnd(&L.flags[b], ~1);     // some vertex has a neighbour: not diagonal
    DSRG_STAMP(5);
#undef DSRG_STAMP
}

// Split build, stage 3: norm = 1/sqrt(K 1 + 1e-20) (pairwise.cpp:44,54-57) through
// Permutohedral::seqCompute (permutohedral.cpp:476-527, blur in double), and the "diagonal" flag.
template <int D, int VPT>
__global__ __launch_bounds__(kWG) void lattice_norm_kernel(LatticeView L) {
    constexpr int D1 = D + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int N = L.N, Mcap = L.Mcap, E = N * D1, M = L.M[b];
    float *val = reinterpret_cast<float *>(smem);                             // [Mcap+1]
    const uint16_t *vid = L.vid + (size_t)b * D1 * N;
    const float *bary = L.bary + (size_t)b * D1 * N;
    const rsrc_t r_nb = make_rsrc(L.nb + (size_t)b * D1 * Mcap, sizeof(uint32_t) * (size_t)D1 * Mcap);
    const rsrc_t r_rs = make_rsrc(L.row_start + (size_t)b * (Mcap + 2), sizeof(uint16_t) * (size_t)(Mcap + 2));
    const rsrc_t r_cw = make_rsrc(L.csr_w + (size_t)b * E, sizeof(float) * (size_t)E);
    uint32_t rs0[VPT], rs1[VPT];
#pragma unroll
    for (int k = 0; k < VPT; k++) {
        rs0[k] = ld_u16(r_rs, (uint32_t)tid * 2u, (uint32_t)k * (kWG * 2u));
        rs1[k] = ld_u16(r_rs, (uint32_t)tid * 2u, (uint32_t)k * (kWG * 2u) + 2u);
    }
    int multi = 0;                                                               // some vertex has several contributors
#pragma unroll
    for (int k = 0; k < VPT; k++) {
        const int v = tid + k * kWG;
        if (v < M) {
            float s = 0.0f;
            for (uint32_t pos = rs0[k]; pos < rs1[k]; pos++) s = s + ld_f32(r_cw, pos * 4u) * 1.0f;
            val[v] = s;
            multi |= (rs1[k] - rs0[k] != 1u);
        }
    }
    if (tid == 0) val[M] = 0.0f;                                                 // zero sentinel = "no neighbour"
    __syncthreads();
    int has_nb = 0;
    for (int j = 0; j <= D; j++) {
        uint32_t word[VPT];
#pragma unroll
        for (int k = 0; k < VPT; k++)
            word[k] = ld_u32(r_nb, (uint32_t)tid * 4u, (uint32_t)j * (uint32_t)Mcap * 4u + (uint32_t)k * (kWG * 4u));
        float nv[VPT];
#pragma unroll
        for (int k = 0; k < VPT; k++) {
            const int v = tid + k * kWG;
            const bool ok = v < M;
            const int n1 = ok ? (int)(word[k] & 0xffffu) : M, n2 = ok ? (int)(word[k] >> 16) : M;
            has_nb |= (n1 != M) | (n2 != M);
            const float s = val[n1] + val[n2];
            nv[k] = (float)((double)val[ok ? v : M] + 0.5 * (double)s);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < VPT; k++) {
            const int v = tid + k * kWG;
            if (v < M) val[v] = nv[k];
        }
        __syncthreads();
    }
    const float alpha = 1.0f / (1.0f + exp2f((float)-D));
    float *norm = L.norm + (size_t)b * N;
    for (int i = tid; i < N; i += kWG) {
        float out = 0.0f;
#pragma unroll
        for (int r = 0; r <= D; r++) {
            float t = bary[(size_t)r * N + i] * val[vid[(size_t)r * N + i]];
            t = t * alpha;
            out = out + t;
        }
        norm[i] = (float)(1.0 / sqrt((double)out + 1e-20));
    }
    const int any_nb = __syncthreads_or(has_nb);
    const int any_multi = __syncthreads_or(multi);
    // bit 2: every vertex has exactly one contributor and vice versa (row v of the splat list is entry v)
    if (tid == 0) L.flags[b] = (L.flags[b] & ~5) | ((!any_nb && M == E) ? 1 : 0) | ((M == E && !any_multi) ? 4 : 0);
}

// ---------------------------------------------------------------------------------
// Pixel-local test of a d = 2 lattice (flag kLatticeLocal) and the per-pixel form the mean-field update kernel evaluates
// (meanfield.hip, GaussLocal).  At training scale the spatial kernel of CRF.py:31-32 has sigma = 3/12 = 0.25 px: a pixel's
// simplex shares no vertex with any other pixel's, so every vertex has one contributor (flag bit 2) and the only blur
// neighbours (permutohedral.cpp:303-318) of its three corners are each other — corner r+1 = corner r + the step of axis
// j(r), i.e. along each axis exactly one pair (a, b) with n2(a) = b and n1(b) = a.  The kernel VERIFIES that structure for
// every pixel from the built tables (it does not assume it), relabels the corners so that axis j pairs relabelled corners
// (j, j+1 mod 3), and stores norm + relabelled weights + the relabelled index of original corner 2.  One workgroup.
__global__ __launch_bounds__(kWG) void lattice_local_kernel(LatticeView L) {
    const int N = L.N, Mcap = L.Mcap, M = L.M[0], tid = threadIdx.x;
    const bool single = (L.flags[0] & 4) != 0;            // every vertex has exactly one contributor
    int bad = single ? 0 : 1;
    for (int i = tid; i < N && single; i += kWG) {
        uint32_t V[3];
#pragma unroll
        for (int r = 0; r < 3; r++) V[r] = L.vid[(size_t)r * N + i];
        auto local_of = [&](uint32_t v) { return v == V[0] ? 0 : v == V[1] ? 1 : v == V[2] ? 2 : -1; };
        int pa[3], pb[3];                                  // the exchanging pair of each axis (original corner indices)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            pa[j] = pb[j] = -1;
            int links = 0;
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const uint32_t w = L.nb[(size_t)j * Mcap + V[r]];
                const uint32_t n1 = w & 0xffffu, n2 = w >> 16;
                if (n2 != (uint32_t)M) {                   // r -> n2 must stay inside the pixel and be mirrored by n1
                    const int t = local_of(n2);
                    if (t < 0 || t == r || (L.nb[(size_t)j * Mcap + n2] & 0xffffu) != V[r]) bad = 1;
                    else { pa[j] = r; pb[j] = t; links++; }
                }
                if (n1 != (uint32_t)M) {
                    const int t = local_of(n1);
                    if (t < 0 || t == r || (L.nb[(size_t)j * Mcap + n1] >> 16) != V[r]) bad = 1;
                }
            }
            if (links != 1) bad = 1;
        }
        if (V[0] == V[1] || V[1] == V[2] || V[0] == V[2]) bad = 1;
        // relabelling sigma: the corner shared by the pairs of axes 2 and 0 -> 0, of axes 0 and 1 -> 1, of axes 1 and 2 -> 2
        int sigma[3] = {-1, -1, -1};
        if (!bad) {
            auto shared = [&](int j0, int j1) {
                if (pa[j0] == pa[j1] || pa[j0] == pb[j1]) return pa[j0];
                if (pb[j0] == pa[j1] || pb[j0] == pb[j1]) return pb[j0];
                return -1;
            };
            const int c0 = shared(2, 0), c1 = shared(0, 1), c2 = shared(1, 2);
            if (c0 < 0 || c1 < 0 || c2 < 0 || c0 == c1 || c1 == c2 || c0 == c2) bad = 1;
            else { sigma[c0] = 0; sigma[c1] = 1; sigma[c2] = 2; }
        }
        float w[3] = {0.f, 0.f, 0.f};
        if (!bad) {
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const float br = L.bary[(size_t)r * N + i];
#pragma unroll
                for (int q = 0; q < 3; q++) if (sigma[r] == q) w[q] = br;
            }
        }
        L.loc_a[(size_t)4 * i + 0] = L.norm[i];
        L.loc_a[(size_t)4 * i + 1] = w[0];
        L.loc_a[(size_t)4 * i + 2] = w[1];
        L.loc_a[(size_t)4 * i + 3] = w[2];
        L.loc_z[i] = bad ? 0u : (uint32_t)sigma[2];
    }
    const int any_bad = __syncthreads_or(bad);
    if (tid == 0) L.flags[0] = (L.flags[0] & ~kLatticeLocal) | (any_bad ? 0 : kLatticeLocal);
}

// ---------------------------------------------------------------------------------
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static void lattice_layout(int d, int N, int nlat, size_t off[22], size_t &total) {
    const int d1 = d + 1, Npad = (N + 3) / 4 * 4, Mcap = Npad * d1, E = N * d1, Epad = Npad * d1;
    const int KW = (d * 16 + 31) / 32;
    size_t sz[22] = {
        sizeof(int) * (size_t)nlat,                          // M
        sizeof(uint16_t) * (size_t)E * nlat,                 // vid
        sizeof(float) * (size_t)E * nlat,                    // bary
        sizeof(uint32_t) * (size_t)d1 * Mcap * nlat,         // nb
        sizeof(uint16_t) * (size_t)(Mcap + 2) * nlat,        // row_start
        sizeof(unsigned long long) * (size_t)Epad * nlat,    // ckeys_e
        sizeof(float) * (size_t)E * nlat,                    // csr_w
        sizeof(float) * (size_t)N * nlat,                    // norm
        sizeof(uint32_t) * (size_t)Epad * KW * nlat,         // key_e
        sizeof(uint16_t) * (size_t)Epad * nlat,              // slot_e
        sizeof(uint32_t) * (size_t)Mcap * KW * nlat,         // key_v
        sizeof(int) * (size_t)nlat,                          // flags
        sizeof(uint16_t) * (size_t)lattice_table_cap(Mcap) * nlat,     // tab_g
        sizeof(unsigned long long) * (size_t)Mcap * nlat,     // ckeys_g
        d == 2 ? sizeof(float) * 4 * (size_t)N * nlat : 0,   // loc_a
        d == 2 ? sizeof(uint32_t) * (size_t)N * nlat : 0,    // loc_z
        sizeof(uint16_t) * (size_t)Mcap * nlat,              // first_pix
        sizeof(float) * (size_t)Mcap * nlat,                 // first_w
        sizeof(uint16_t) * (size_t)E * nlat,                 // x_pix
        sizeof(float) * (size_t)E * nlat,                    // x_w
        sizeof(int) * (size_t)nlat,                          // nextra
        sizeof(int) * 32 * (size_t)nlat,                     // embed_bad
    };
    size_t cur = 0;
    for (int i = 0; i < 22; i++) { off[i] = cur; cur += align_up(sz[i], 256); }
    total = cur;
}

size_t lattice_bytes(int d, int N, int nlat) {
    size_t off[22], total;
    lattice_layout(d, N, nlat, off, total);
    return total;
}

void lattice_carve(LatticeView &L, void *base, int d, int N, int nlat) {
    size_t off[22], total;
    lattice_layout(d, N, nlat, off, total);
    unsigned char *p = static_cast<unsigned char *>(base);
    L.d = d; L.N = N; L.Mcap = ((N + 3) / 4 * 4) * (d + 1); L.nlat = nlat;
    L.M = reinterpret_cast<int *>(p + off[0]);
    L.vid = reinterpret_cast<uint16_t *>(p + off[1]);
    L.bary = reinterpret_cast<float *>(p + off[2]);
    L.nb = reinterpret_cast<uint32_t *>(p + off[3]);
    L.row_start = reinterpret_cast<uint16_t *>(p + off[4]);
    L.csr_w = reinterpret_cast<float *>(p + off[6]);
    L.norm = reinterpret_cast<float *>(p + off[7]);
    L.key_e = reinterpret_cast<uint32_t *>(p + off[8]);
    L.slot_e = reinterpret_cast<uint16_t *>(p + off[9]);
    L.ckeys_e = reinterpret_cast<unsigned long long *>(p + off[5]);
    L.embed_bad = reinterpret_cast<int *>(p + off[21]);
    L.key_v = reinterpret_cast<uint32_t *>(p + off[10]);
    L.flags = reinterpret_cast<int *>(p + off[11]);
    L.tab_g = reinterpret_cast<uint32_t *>(p + off[12]);
    L.ckeys_g = reinterpret_cast<unsigned long long *>(p + off[13]);
    L.loc_a = d == 2 ? reinterpret_cast<float *>(p + off[14]) : nullptr;
    L.loc_z = d == 2 ? reinterpret_cast<uint32_t *>(p + off[15]) : nullptr;
    L.first_pix = reinterpret_cast<uint16_t *>(p + off[16]);
    L.first_w = reinterpret_cast<float *>(p + off[17]);
    L.x_pix = reinterpret_cast<uint16_t *>(p + off[18]);
    L.x_w = reinterpret_cast<float *>(p + off[19]);
    L.nextra = reinterpret_cast<int *>(p + off[20]);
}

void lattice_feat_init(LatticeFeat &F, int d, int W, int H, float sx, float sy, float sr, float sg, float sb) {
    F.sx = sx; F.sy = sy; F.sr = sr; F.sg = sg; F.sb = sb; F.W = W; F.H = H;
    // permutohedral.cpp:179-182, evaluated in double exactly like the reference
    const float inv_std_dev = (float)(sqrt(2.0 / 3.0) * (double)(d + 1));
    for (int i = 0; i < 5; i++) F.scale[i] = 0.0f;
    for (int i = 0; i < d; i++) F.scale[i] = (float)(1.0 / sqrt((double)((i + 2) * (i + 1))) * (double)inv_std_dev);
}

static size_t build_lds_bytes(int d, int N, bool with_wl = false, bool with_keys = false) {
    const int d1 = d + 1, Npad = (N + 3) / 4 * 4, Mcap = Npad * d1, E = N * d1, Epad = Npad * d1;
    const int cap = lattice_table_cap(Mcap);
    size_t a = align_up((size_t)cap * 2 + 32 * 4 + 2 * (size_t)(Epad / 32 + 1) * 4, 16);
    if (with_keys) a += align_up((size_t)Mcap * (d == 5 ? 8 : 4), 16);
    size_t b = align_up((size_t)(Mcap + 1) * 4, 16) + align_up((size_t)E * 2, 16) + 32 * 4;
    if (with_wl) b += (size_t)E * 4;
    return a > b ? a : b;
}

bool lattice_supported(int d, int N) {
    if (d != 2 && d != 5) return false;
    if (N < 1) return false;
    const int Mcap = ((N + 3) / 4 * 4) * (d + 1);
    if (Mcap + 1 >= 65536) return false;                         // uint16 ids + sentinel
    if (Mcap > kBuildVPT * kWG) return false;                    // register Jacobi bound
    const int cap = lattice_table_cap(Mcap);
    if ((double)Mcap > 0.8 * (double)cap) return false;          // linear probing load factor
    if (build_lds_bytes(d, N) > 158 * 1024) return false;    // leave room for the static word of __syncthreads_or
    return true;
}

int launch_lattice_build(const LatticeView &L, const LatticeFeat &F, const LatticeColours &col, int nlat,
                         hipStream_t stream) {
    if (!lattice_supported(L.d, L.N))
        return set_error(DSRG_ERR_UNSUPPORTED,
                         "lattice with d=%d over %d pixels does not fit the LDS-resident path", L.d, L.N);
    const int cap = lattice_table_cap(L.Mcap);
    const bool lds_keys = build_lds_bytes(L.d, L.N, false, true) <= 150 * 1024;
    const bool wl_in_lds = build_lds_bytes(L.d, L.N, true, lds_keys) <= 150 * 1024;
    const size_t lds = build_lds_bytes(L.d, L.N, wl_in_lds, lds_keys);
    const int vpt = (L.Mcap + kWG - 1) / kWG;
    unsigned long long *dbg = L.d == 5 ? reinterpret_cast<unsigned long long *>(g_build_dbg) : nullptr;
    // split the build over more workgroups (a second launch of nsplit neighbour-search and kCsrParts CSR workgroups per
    // lattice) when the hash table and the compact keys fit one workgroup's LDS next to each other; a CSR workgroup needs
    // the row counters [Mcap+1], the sorted entries [E] and the scan scratch
    const size_t csr_lds = align_up((size_t)(L.Mcap + 1) * 4, 16) + align_up((size_t)L.N * (L.d + 1) * 2, 16) + 32 * 4;
    const size_t search_lds = (size_t)cap * 2 + (lds_keys ? (((size_t)L.Mcap * (L.d == 5 ? 8 : 4) + 15) & ~(size_t)15) : 0);
    const size_t neigh_lds = search_lds > csr_lds ? search_lds : csr_lds;
    const int split = (neigh_lds <= 150 * 1024 && L.Mcap <= 16 * kWG) ? 1 : 0;
    // at least ceil(Mcap / 1024) search workgroups so that no thread gets a second vertex; beyond that as many as keep the
    // launch one round of workgroups on the chip (hashing and probing 12 neighbour keys is ~7 us of dependent work per wave)
    int nsplit = vpt < 8 ? 8 : vpt;
    if (240 / nlat - kCsrParts > nsplit) nsplit = 240 / nlat - kCsrParts < 32 ? 240 / nlat - kCsrParts : 32;
    {
        const int Npad = (L.N + 3) & ~3;
        const dim3 grid((Npad + kEmbedWG - 1) / kEmbedWG, nlat);
        if (grid.x > 32) return set_error(DSRG_ERR_UNSUPPORTED, "map of %d pixels exceeds the embedding kernel's flag array", L.N);
        if (L.d == 5 && !col.im_u8 && !(col.images && col.im_out))
            return set_error(DSRG_ERR_INVALID, "bilateral lattices need an image");
        if (L.d == 2) hipLaunchKernelGGL(lattice_embed_kernel<2>, grid, dim3(kEmbedWG), 0, stream, L, F, col);
        else hipLaunchKernelGGL(lattice_embed_kernel<5>, grid, dim3(kEmbedWG), 0, stream, L, F, col);
        DSRG_LAUNCH_CHECK();
    }
#define DSRG_BUILD(D_, V_)                                                                                    \
    do {                                                                                                      \
        static LdsGrant granted, granted_n, granted_m;                                                         \
        int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&lattice_build_kernel<D_, V_>), lds, granted); \
        if (rc) return rc;                                                                                    \
        hipLaunchKernelGGL((lattice_build_kernel<D_, V_>), dim3(nlat), dim3(kWG), lds, stream, L, cap,         \
                           (int)wl_in_lds, (int)lds_keys, split, dbg);                                        \
        if (split) {                                                                                          \
            rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&lattice_neigh_kernel<D_, V_>), neigh_lds, granted_n); \
            if (rc) return rc;                                                                                \
            hipLaunchKernelGGL((lattice_neigh_kernel<D_, V_>), dim3(nlat * (nsplit + kCsrParts)), dim3(kWG), neigh_lds, stream, L, \
                               cap, (int)lds_keys, nsplit, dbg);                                              \
            if (D_ == 5) {   /* the filter kernel over a plane of ones (meanfield.hip) */                      \
                rc = launch_lattice_norm_pass(L, nlat, stream);                                               \
                if (rc) return rc;                                                                            \
            } else {                                                                                          \
                const size_t norm_lds = (size_t)(L.Mcap + 1) * 4;                                             \
                rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&lattice_norm_kernel<D_, V_>), norm_lds, granted_m); \
                if (rc) return rc;                                                                            \
                hipLaunchKernelGGL((lattice_norm_kernel<D_, V_>), dim3(nlat), dim3(kWG), norm_lds, stream, L); \
            }                                                                                                 \
        }                                                                                                     \
    } while (0)
#define DSRG_BUILD_V(D_)                                                                                      \
    do {                                                                                                      \
        if (vpt <= 4) DSRG_BUILD(D_, 4); else if (vpt <= 10) DSRG_BUILD(D_, 10);                              \
        else if (vpt <= 16) DSRG_BUILD(D_, 16); else if (vpt <= 25) DSRG_BUILD(D_, 25); else DSRG_BUILD(D_, 32); \
    } while (0)
    if (L.d == 2) DSRG_BUILD_V(2); else DSRG_BUILD_V(5);
#undef DSRG_BUILD_V
#undef DSRG_BUILD
    DSRG_LAUNCH_CHECK();
    if (L.d == 2 && nlat == 1) {
        hipLaunchKernelGGL(lattice_local_kernel, dim3(1), dim3(kWG), 0, stream, L);
        DSRG_LAUNCH_CHECK();
    }
    return DSRG_OK;
}

}  // namespace dsrg
