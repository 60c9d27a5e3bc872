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

namespace dsrg {

template <int D> struct KeyWords { static constexpr int value = (D * 16 + 31) / 32; };

template <int KW> __device__ __forceinline__ uint32_t hash_key(const uint32_t (&w)[KW]) {
    uint32_t h = 0;
#pragma unroll
    for (int i = 0; i < KW; i++) h = (h ^ w[i]) * 0x9E3779B1u;
    return h ^ (h >> 13);
}
template <int KW> __device__ __forceinline__ void load_key(uint32_t (&w)[KW], const uint32_t *p) {
#pragma unroll
    for (int i = 0; i < KW; i++) w[i] = p[i];
}
template <int KW> __device__ __forceinline__ bool key_eq(const uint32_t (&a)[KW], const uint32_t *p) {
    bool eq = true;
#pragma unroll
    for (int i = 0; i < KW; i++) eq &= (a[i] == p[i]);
    return eq;
}
template <int D> __device__ __forceinline__ void pack_key(uint32_t (&w)[KeyWords<D>::value], const short (&k)[D]) {
#pragma unroll
    for (int i = 0; i < KeyWords<D>::value; i++) w[i] = 0;
#pragma unroll
    for (int i = 0; i < D; i++) w[i >> 1] |= (uint32_t)(uint16_t)k[i] << ((i & 1) * 16);
}
template <int D> __device__ __forceinline__ void unpack_key(short (&k)[D], const uint32_t (&w)[KeyWords<D>::value]) {
#pragma unroll
    for (int i = 0; i < D; i++) k[i] = (short)(uint16_t)(w[i >> 1] >> ((i & 1) * 16));
}

// exclusive scan of one int per thread over the workgroup; `scratch` holds >= 17 ints.
// returns the exclusive prefix; *total receives the workgroup sum.
__device__ __forceinline__ int block_exclusive_scan(int x, int *scratch, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    int incl = x;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int y = __shfl_up(incl, off, 64);
        if (lane >= off) incl += y;
    }
    __syncthreads();
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int w = 0; w < nwaves; w++) { int t = scratch[w]; scratch[w] = run; run += t; }
        scratch[16] = run;
    }
    __syncthreads();
    *total = scratch[16];
    return scratch[wave] + incl - x;
}

__host__ __device__ inline int lattice_table_cap(int Mcap) {
    int cap = 1024;
    while (cap < 2 * Mcap && cap < 32768) cap <<= 1;
    return cap;
}

constexpr int kBuildVPT = 32;   // register-Jacobi bound of the norm pass: Mcap <= 32*1024

template <int D>
__global__ __launch_bounds__(kWG) void lattice_build_kernel(LatticeView L, LatticeFeat F,
                                                              const unsigned char *__restrict__ im, int cap) {
    constexpr int D1 = D + 1, KW = KeyWords<D>::value;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int N = L.N, Npad = (N + 3) & ~3, E = N * D1, Epad = Npad * D1, Mcap = L.Mcap;
    const uint32_t mask = (uint32_t)cap - 1u;

    int *table = reinterpret_cast<int *>(smem);                   // [cap]
    int *scan_scratch = reinterpret_cast<int *>(smem + (size_t)cap * 4);   // [32]

    uint16_t *vid = L.vid + (size_t)b * D1 * N;
    float *bary = L.bary + (size_t)b * D1 * N;
    uint32_t *nb = L.nb + (size_t)b * D1 * Mcap;
    uint32_t *row_start = L.row_start + (size_t)b * (Mcap + 1);
    uint16_t *csr_pix = L.csr_pix + (size_t)b * E;
    float *csr_w = L.csr_w + (size_t)b * E;
    float *norm = L.norm + (size_t)b * N;
    uint32_t *key_e = L.key_e + (size_t)b * Epad * KW;
    uint16_t *slot_e = L.slot_e + (size_t)b * Epad;
    uint32_t *key_v = L.key_v + (size_t)b * Mcap * KW;

    for (int h = tid; h < cap; h += kWG) table[h] = -1;

    // ---- phase 1: embed every pixel (incl. the SSE zero padding) -> keys + weights
    const float invdplus1 = 1.0f / (float)D1;      // permutohedral.cpp:148
    const float dplus1 = (float)D1;                // :149
    for (int i = tid; i < Npad; i += kWG) {
        float f[D];
#pragma unroll
        for (int j = 0; j < D; j++) f[j] = 0.0f;
        if (i < N) {
            const int x = i % F.W, y = i / F.W;     // densecrf.cpp:63-67,72-79
            f[0] = (float)x / F.sx;
            f[1] = (float)y / F.sy;
            if constexpr (D == 5) {
                const unsigned char *px = im + ((size_t)b * N + i) * 3;
                f[2] = (float)px[0] / F.sr;
                f[3] = (float)px[1] / F.sg;
                f[4] = (float)px[2] / F.sb;
            }
        }
        float elevated[D1], rem0[D1], rank[D1];
        float sm = 0.0f;                            // :201-207
#pragma unroll
        for (int j = D; j > 0; j--) {
            float cf = f[j - 1] * F.scale[j - 1];
            float jc = (float)j * cf;
            elevated[j] = sm - jc;
            sm = sm + cf;
        }
        elevated[0] = sm;
        float sum = 0.0f;                           // :210-220
#pragma unroll
        for (int k = 0; k <= D; k++) {
            float v = rintf(invdplus1 * elevated[k]);   // round-half-even, as _mm_cvtps_epi32
            rem0[k] = v * dplus1;
            sum = sum + v;
            rank[k] = 0.0f;
        }
#pragma unroll
        for (int a = 0; a < D; a++) {               // :225-233
            float di = elevated[a] - rem0[a];
#pragma unroll
            for (int c = a + 1; c <= D; c++) {
                float dj = elevated[c] - rem0[c];
                float lt = (di < dj) ? 1.0f : 0.0f;
                rank[a] = rank[a] + lt;
                rank[c] = rank[c] + (1.0f - lt);
            }
        }
#pragma unroll
        for (int k = 0; k <= D; k++) {              // :236-242
            rank[k] = rank[k] + sum;
            float add = (rank[k] < 0.0f) ? dplus1 : 0.0f;
            float sub = (rank[k] >= dplus1) ? dplus1 : 0.0f;
            float as = add - sub;
            rank[k] = rank[k] + as;
            rem0[k] = rem0[k] + as;
        }
        float bc[D + 2];                            // :245-258
#pragma unroll
        for (int q = 0; q < D + 2; q++) bc[q] = 0.0f;
#pragma unroll
        for (int k = 0; k <= D; k++) {
            float v = (elevated[k] - rem0[k]) * invdplus1;
            int p = (int)((float)D - rank[k]);
#pragma unroll
            for (int q = 0; q < D + 2; q++) {       // static indexing keeps bc[] in registers
                if (q == p) bc[q] = bc[q] + v;
                if (q == p + 1) bc[q] = bc[q] - v;
            }
        }
        bc[0] = bc[0] + (1.0f + bc[D + 1]);         // :263
#pragma unroll
        for (int r = 0; r <= D; r++) {              // :268-275
            short key[D];
#pragma unroll
            for (int k = 0; k < D; k++) {
                int rk = (int)rank[k];
                int canon = (rk <= D - r) ? r : r - D1;    // canonical[r][rk], :171-176
                key[k] = (short)(int)(rem0[k] + (float)canon);
            }
            uint32_t w[KW];
            pack_key<D>(w, key);
#pragma unroll
            for (int q = 0; q < KW; q++) key_e[((size_t)i * D1 + r) * KW + q] = w[q];
            if (i < N) bary[(size_t)r * N + i] = bc[r];
        }
    }
    __syncthreads();

    // ---- phase 2: deduplicate keys (open addressing, linear probing, LDS CAS)
    for (int e = tid; e < Epad; e += kWG) {
        uint32_t w[KW];
        load_key<KW>(w, key_e + (size_t)e * KW);
        uint32_t h = hash_key<KW>(w) & mask;
        for (;;) {
            int old = atomicCAS(&table[h], -1, e);
            if (old == -1) break;
            if (key_eq<KW>(w, key_e + (size_t)old * KW)) break;
            h = (h + 1) & mask;
        }
        slot_e[e] = (uint16_t)h;
    }
    __syncthreads();

    // ---- phase 3: dense vertex ids in slot order; table[h] becomes the id
    const int chunk = cap / kWG;
    int occupied = 0;
    for (int q = 0; q < chunk; q++) occupied += (table[tid * chunk + q] >= 0);
    int M;
    int base = block_exclusive_scan(occupied, scan_scratch, &M);
    for (int q = 0; q < chunk; q++) {
        const int h = tid * chunk + q;
        const int rep = table[h];
        if (rep >= 0) {
#pragma unroll
            for (int t = 0; t < KW; t++) key_v[(size_t)base * KW + t] = key_e[(size_t)rep * KW + t];
            table[h] = base++;
        }
    }
    if (tid == 0) L.M[b] = M;
    __syncthreads();

    // ---- phase 4: vertex id of every real (pixel, corner) entry
    for (int e = tid; e < E; e += kWG) {
        const int i = e / D1, r = e - i * D1;
        vid[(size_t)r * N + i] = (uint16_t)table[slot_e[e]];
    }

    // ---- phase 5: blur neighbours (permutohedral.cpp:303-318)
    for (int v = tid; v < M; v += kWG) {
        uint32_t w[KW];
        load_key<KW>(w, key_v + (size_t)v * KW);
        short k0[D];
        unpack_key<D>(k0, w);
#pragma unroll
        for (int j = 0; j <= D; j++) {
            short n1[D], n2[D];
#pragma unroll
            for (int k = 0; k < D; k++) { n1[k] = (short)(k0[k] - 1); n2[k] = (short)(k0[k] + 1); }
            if (j < D) { n1[j < D ? j : 0] = (short)(k0[j < D ? j : 0] + D); n2[j < D ? j : 0] = (short)(k0[j < D ? j : 0] - D); }
            uint32_t res[2];
#pragma unroll
            for (int s = 0; s < 2; s++) {
                uint32_t q[KW];
                if (s == 0) pack_key<D>(q, n1); else pack_key<D>(q, n2);
                uint32_t h = hash_key<KW>(q) & mask;
                uint32_t found = (uint32_t)Mcap;
                for (;;) {
                    int t = table[h];
                    if (t < 0) break;
                    if (key_eq<KW>(q, key_v + (size_t)t * KW)) { found = (uint32_t)t; break; }
                    h = (h + 1) & mask;
                }
                res[s] = found;
            }
            nb[(size_t)j * Mcap + v] = res[0] | (res[1] << 16);
        }
    }
    __syncthreads();

    // ---- phase 6: CSR of the splat, contributions in reference order (entry index ascending)
    uint32_t *cnt = reinterpret_cast<uint32_t *>(smem);                       // [Mcap+1]
    uint16_t *csr_e = reinterpret_cast<uint16_t *>(smem + (((size_t)(Mcap + 1) * 4 + 15) & ~(size_t)15));   // [E]
    int *scan2 = reinterpret_cast<int *>(reinterpret_cast<unsigned char *>(csr_e) + (((size_t)E * 2 + 15) & ~(size_t)15));
    for (int v = tid; v <= Mcap; v += kWG) cnt[v] = 0;
    __syncthreads();
    for (int e = tid; e < E; e += kWG) {
        const int i = e / D1, r = e - i * D1;
        atomicAdd(&cnt[vid[(size_t)r * N + i]], 1u);
    }
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
            row_start[v] = (uint32_t)run;
            run += c;
        }
        if (tid == 0) row_start[M] = (uint32_t)E;
    }
    __syncthreads();
    for (int e = tid; e < E; e += kWG) {
        const int i = e / D1, r = e - i * D1;
        uint32_t pos = atomicAdd(&cnt[vid[(size_t)r * N + i]], 1u);
        csr_e[pos] = (uint16_t)e;
    }
    __syncthreads();
    for (int v = tid; v < M; v += kWG) {          // cnt[v] is now the END of segment v
        const int s = v == 0 ? 0 : (int)cnt[v - 1], t = (int)cnt[v];
        for (int a = s + 1; a < t; a++) {         // insertion sort, segments are short
            uint16_t x = csr_e[a];
            int c = a - 1;
            while (c >= s && csr_e[c] > x) { csr_e[c + 1] = csr_e[c]; c--; }
            csr_e[c + 1] = x;
        }
    }
    __syncthreads();
    for (int pos = tid; pos < E; pos += kWG) {
        const int e = csr_e[pos];
        const int i = e / D1, r = e - i * D1;
        csr_pix[pos] = (uint16_t)i;
        csr_w[pos] = bary[(size_t)r * N + i];
    }
    __syncthreads();

    // ---- phase 7: norm = 1/sqrt(K 1 + 1e-20)  (pairwise.cpp:44,54-57), one channel through
    // Permutohedral::seqCompute (permutohedral.cpp:476-527): blur evaluated in double
    float *val = reinterpret_cast<float *>(smem);                             // [Mcap+1]
    for (int v = tid; v < M; v += kWG) {
        float s = 0.0f;
        const uint32_t a = row_start[v], z = row_start[v + 1];
        for (uint32_t pos = a; pos < z; pos++) s = s + csr_w[pos] * 1.0f;
        val[v] = s;
    }
    if (tid == 0) val[Mcap] = 0.0f;
    __syncthreads();
    for (int j = 0; j <= D; j++) {
        float nv[kBuildVPT];
#pragma unroll
        for (int k = 0; k < kBuildVPT; k++) {
            const int v = tid + k * kWG;
            if (v < M) {
                const uint32_t t = nb[(size_t)j * Mcap + v];
                const float s = val[t & 0xffffu] + val[t >> 16];
                nv[k] = (float)((double)val[v] + 0.5 * (double)s);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kBuildVPT; k++) {
            const int v = tid + k * kWG;
            if (v < M) val[v] = nv[k];
        }
        __syncthreads();
    }
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
}

// ---------------------------------------------------------------------------------
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static void lattice_layout(int d, int N, int nlat, size_t off[12], size_t &total) {
    const int d1 = d + 1, Npad = (N + 3) / 4 * 4, Mcap = Npad * d1, E = N * d1, Epad = Npad * d1;
    const int KW = (d * 16 + 31) / 32;
    size_t sz[11] = {
        sizeof(int) * (size_t)nlat,                          // M
        sizeof(uint16_t) * (size_t)E * nlat,                 // vid
        sizeof(float) * (size_t)E * nlat,                    // bary
        sizeof(uint32_t) * (size_t)d1 * Mcap * nlat,         // nb
        sizeof(uint32_t) * (size_t)(Mcap + 1) * nlat,        // row_start
        sizeof(uint16_t) * (size_t)E * nlat,                 // csr_pix
        sizeof(float) * (size_t)E * nlat,                    // csr_w
        sizeof(float) * (size_t)N * nlat,                    // norm
        sizeof(uint32_t) * (size_t)Epad * KW * nlat,         // key_e
        sizeof(uint16_t) * (size_t)Epad * nlat,              // slot_e
        sizeof(uint32_t) * (size_t)Mcap * KW * nlat,         // key_v
    };
    size_t cur = 0;
    for (int i = 0; i < 11; i++) { off[i] = cur; cur += align_up(sz[i], 256); }
    total = cur;
}

size_t lattice_bytes(int d, int N, int nlat) {
    size_t off[12], total;
    lattice_layout(d, N, nlat, off, total);
    return total;
}

void lattice_carve(LatticeView &L, void *base, int d, int N, int nlat) {
    size_t off[12], total;
    lattice_layout(d, N, nlat, off, total);
    unsigned char *p = static_cast<unsigned char *>(base);
    L.d = d; L.N = N; L.Mcap = ((N + 3) / 4 * 4) * (d + 1); L.nlat = nlat;
    L.M = reinterpret_cast<int *>(p + off[0]);
    L.vid = reinterpret_cast<uint16_t *>(p + off[1]);
    L.bary = reinterpret_cast<float *>(p + off[2]);
    L.nb = reinterpret_cast<uint32_t *>(p + off[3]);
    L.row_start = reinterpret_cast<uint32_t *>(p + off[4]);
    L.csr_pix = reinterpret_cast<uint16_t *>(p + off[5]);
    L.csr_w = reinterpret_cast<float *>(p + off[6]);
    L.norm = reinterpret_cast<float *>(p + off[7]);
    L.key_e = reinterpret_cast<uint32_t *>(p + off[8]);
    L.slot_e = reinterpret_cast<uint16_t *>(p + off[9]);
    L.key_v = reinterpret_cast<uint32_t *>(p + off[10]);
}

void lattice_feat_init(LatticeFeat &F, int d, int W, int H, float sx, float sy, float sr, float sg, float sb) {
    F.sx = sx; F.sy = sy; F.sr = sr; F.sg = sg; F.sb = sb; F.W = W; F.H = H;
    // permutohedral.cpp:179-182, evaluated in double exactly like the reference
    const float inv_std_dev = (float)(sqrt(2.0 / 3.0) * (double)(d + 1));
    for (int i = 0; i < 5; i++) F.scale[i] = 0.0f;
    for (int i = 0; i < d; i++) F.scale[i] = (float)(1.0 / sqrt((double)((i + 2) * (i + 1))) * (double)inv_std_dev);
}

static size_t build_lds_bytes(int d, int N) {
    const int d1 = d + 1, Npad = (N + 3) / 4 * 4, Mcap = Npad * d1, E = N * d1;
    const int cap = lattice_table_cap(Mcap);
    size_t a = (size_t)cap * 4 + 32 * 4;
    size_t b = align_up((size_t)(Mcap + 1) * 4, 16) + align_up((size_t)E * 2, 16) + 32 * 4;
    return a > b ? a : b;
}

bool lattice_supported(int d, int N) {
    if (d != 2 && d != 5) return false;
    if (N < 1) return false;
    const int Mcap = ((N + 3) / 4 * 4) * (d + 1);
    if (Mcap + 1 >= 65536) return false;                         // uint16 ids + sentinel
    if (Mcap > kBuildVPT * kWG) return false;                    // register Jacobi bound
    const int cap = lattice_table_cap(Mcap);
    if ((double)Mcap > 0.9 * (double)cap) return false;          // linear probing load factor
    if (build_lds_bytes(d, N) > 160 * 1024) return false;
    return true;
}

int launch_lattice_build(const LatticeView &L, const LatticeFeat &F, const unsigned char *im, int nlat,
                         hipStream_t stream) {
    if (!lattice_supported(L.d, L.N))
        return set_error(DSRG_ERR_UNSUPPORTED,
                         "lattice with d=%d over %d pixels does not fit the LDS-resident path", L.d, L.N);
    const int cap = lattice_table_cap(L.Mcap);
    const size_t lds = build_lds_bytes(L.d, L.N);
    if (L.d == 2) {
        static size_t granted2 = 0;
        int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&lattice_build_kernel<2>), lds, granted2);
        if (rc) return rc;
        hipLaunchKernelGGL(lattice_build_kernel<2>, dim3(nlat), dim3(kWG), lds, stream, L, F, im, cap);
    } else {
        static size_t granted5 = 0;
        int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&lattice_build_kernel<5>), lds, granted5);
        if (rc) return rc;
        hipLaunchKernelGGL(lattice_build_kernel<5>, dim3(nlat), dim3(kWG), lds, stream, L, F, im, cap);
    }
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

}  // namespace dsrg
