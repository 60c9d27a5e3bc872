// Shared device code of the permutohedral lattice construction (small LDS-resident path in
// lattice.hip, large global-memory path in lattice_large.hip): key packing, hashing, the pixel
// embedding of Permutohedral::init, and a workgroup scan.
#pragma once
#include <math.h>
#include "common.h"

namespace dsrg {

// zoom(order=1) of a mean-subtracted (B,3,Hi,Wi) image to the (H,W) map with the (in-1)/(out-1) mapping, + mean pixel,
// np.round, astype(ubyte) (pylayers.py:70-75, CRF.py:32): the three channels of map pixel p of image b, packed r | g<<8 | b<<16
__device__ __forceinline__ uint32_t map_pixel_rgb(const float *__restrict__ images, int b, int Hi, int Wi, int H, int W, int p) {
    const int y = p / W, x = p - y * W;
    const double mean_pixel[3] = {104.0, 117.0, 123.0};
    const double sy = H > 1 ? (double)y * (double)(Hi - 1) / (double)(H - 1) : 0.0;
    const double sx = W > 1 ? (double)x * (double)(Wi - 1) / (double)(W - 1) : 0.0;
    const int y0 = (int)floor(sy), x0 = (int)floor(sx);
    const double fy = sy - y0, fx = sx - x0;
    const int y1 = y0 + 1 < Hi ? y0 + 1 : y0, x1 = x0 + 1 < Wi ? x0 + 1 : x0;
    uint32_t out = 0;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const float *pl = images + ((size_t)b * 3 + ch) * Hi * Wi;
        double v;
        if (fy == 0.0 && fx == 0.0) v = (double)pl[(size_t)y0 * Wi + x0];
        else {
            const double a = (1.0 - fy) * (double)pl[(size_t)y0 * Wi + x0] + fy * (double)pl[(size_t)y1 * Wi + x0];
            const double c = (1.0 - fy) * (double)pl[(size_t)y0 * Wi + x1] + fy * (double)pl[(size_t)y1 * Wi + x1];
            v = (1.0 - fx) * a + fx * c;
        }
        const float vf = (float)v;
        const double r = rint((double)vf + mean_pixel[ch]);      // half-even, like np.round
        out |= (uint32_t)(unsigned char)(long long)r << (8 * ch);
    }
    return out;
}

template <int D> struct KeyWords { static constexpr int value = (D * 16 + 31) / 32; };

template <int KW> __device__ __forceinline__ uint32_t hash_key(const uint32_t (&w)[KW]) {
    uint32_t h = 0;
#pragma unroll
    for (int i = 0; i < KW; i++) { h = (h ^ w[i]) * 0x9E3779B1u; h ^= h >> 15; }
    h *= 0x85EBCA6Bu;                    // avalanche: lattice keys are highly regular
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    return h ^ (h >> 16);
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
// Keys are `short` coordinates (permutohedral.cpp:168,270) packed two per 32-bit word, each biased
// by 0x8000 (an injective re-coding: only key EQUALITY matters).  The bias keeps every field away
// from 0 / 0xFFFF for any realistic lattice, so "all coordinates -1" is one subtraction per word
// with no borrow between fields; a coordinate that would borrow wraps exactly like the short does
// only when the neighbouring field is unaffected — guarded by the range flag computed in phase 1.
template <int D> __device__ __forceinline__ void pack_key(uint32_t (&w)[KeyWords<D>::value], const short (&k)[D]) {
#pragma unroll
    for (int i = 0; i < KeyWords<D>::value; i++) w[i] = 0;
#pragma unroll
    for (int i = 0; i < D; i++) w[i >> 1] |= (uint32_t)(uint16_t)((int)k[i] + 0x8000) << ((i & 1) * 16);
}
// neighbour keys of permutohedral.cpp:307-313 on the packed form:
//   n1 = key - 1 on every stored coordinate, then coordinate j: key[j] + d   (i.e. -1 + (d+1))
//   n2 = key + 1 on every stored coordinate, then coordinate j: key[j] - d   (i.e. +1 - (d+1))
template <int D> __device__ __forceinline__ void neighbour_key(uint32_t (&n)[KeyWords<D>::value],
                                                               const uint32_t (&w)[KeyWords<D>::value], int j, bool plus) {
    constexpr int KW = KeyWords<D>::value;
#pragma unroll
    for (int i = 0; i < KW; i++) {
        const uint32_t ones = (2 * i + 1 < D) ? 0x00010001u : 0x00000001u;      // fields present in word i
        n[i] = plus ? w[i] + ones : w[i] - ones;
    }
    if (j < D) {
        const uint32_t delta = (uint32_t)(D + 1) << ((j & 1) * 16);
#pragma unroll
        for (int i = 0; i < KW; i++)
            if (i == (j >> 1)) n[i] = plus ? n[i] - delta : n[i] + delta;
    }
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


// Embed pixel i (or an SSE zero-padding pixel when i >= N) into the lattice: the d+1 packed vertex keys
// of its simplex and their barycentric weights.  Follows Permutohedral::init, SSE variant
// (CRF/src/permutohedral.cpp:191-275) in the reference's fp32 operation order; features per
// DenseCRF2D::addPairwiseGaussian/Bilateral (CRF/src/densecrf.cpp:61-81).  im: (N,3) uint8 of this image.
// returns bit 0: a coordinate beyond +-32000, bit 1: a coordinate beyond the 12-bit compact range.
// (the colour of pixel i is handed in: callers fetch it ahead of time)
template <int D>
__device__ __forceinline__ int embed_pixel_rgb(const LatticeFeat &F, int i, int N, float pr, float pg, float pb,
                                               uint32_t (&keys)[D + 1][KeyWords<D>::value], float (&bc_out)[D + 1]) {
    constexpr int D1 = D + 1;
    const float invdplus1 = 1.0f / (float)D1;      // permutohedral.cpp:148
    const float dplus1 = (float)D1;                // :149
    float f[D];
#pragma unroll
    for (int j = 0; j < D; j++) f[j] = 0.0f;
    if (i < N) {
        const int x = i % F.W, y = i / F.W;         // densecrf.cpp:63-67,72-79
        f[0] = (float)x / F.sx;
        f[1] = (float)y / F.sy;
        if constexpr (D == 5) {
            f[2] = pr / F.sr;
            f[3] = pg / F.sg;
            f[4] = pb / F.sb;
        }
    }
    float elevated[D1], rem0[D1], rank[D1];
    float sm = 0.0f;                                // :201-207
#pragma unroll
    for (int j = D; j > 0; j--) {
        float cf = f[j - 1] * F.scale[j - 1];
        float jc = (float)j * cf;
        elevated[j] = sm - jc;
        sm = sm + cf;
    }
    elevated[0] = sm;
    float sum = 0.0f;                               // :210-220
#pragma unroll
    for (int k = 0; k <= D; k++) {
        float v = rintf(invdplus1 * elevated[k]);   // round-half-even, as _mm_cvtps_epi32
        rem0[k] = v * dplus1;
        sum = sum + v;
        rank[k] = 0.0f;
    }
#pragma unroll
    for (int a = 0; a < D; a++) {                   // :225-233
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
    for (int k = 0; k <= D; k++) {                  // :236-242
        rank[k] = rank[k] + sum;
        float add = (rank[k] < 0.0f) ? dplus1 : 0.0f;
        float sub = (rank[k] >= dplus1) ? dplus1 : 0.0f;
        float as = add - sub;
        rank[k] = rank[k] + as;
        rem0[k] = rem0[k] + as;
    }
    float bc[D + 2];                                // :245-258
#pragma unroll
    for (int q = 0; q < D + 2; q++) bc[q] = 0.0f;
#pragma unroll
    for (int k = 0; k <= D; k++) {
        float v = (elevated[k] - rem0[k]) * invdplus1;
        int p = (int)((float)D - rank[k]);
#pragma unroll
        for (int q = 0; q < D + 2; q++) {           // static indexing keeps bc[] in registers
            if (q == p) bc[q] = bc[q] + v;
            if (q == p + 1) bc[q] = bc[q] - v;
        }
    }
    bc[0] = bc[0] + (1.0f + bc[D + 1]);             // :263
    int bad = 0;
#pragma unroll
    for (int r = 0; r <= D; r++) {                  // :268-275
        short key[D];
#pragma unroll
        for (int k = 0; k < D; k++) {
            int rk = (int)rank[k];
            int canon = (rk <= D - r) ? r : r - D1;        // canonical[r][rk], :171-176
            key[k] = (short)(int)(rem0[k] + (float)canon);
            bad |= ((key[k] > 32000) | (key[k] < -32000)) ? 1 : 0;
            bad |= ((key[k] >= 2048) | (key[k] < -2048)) ? 2 : 0;
        }
        pack_key<D>(keys[r], key);
        bc_out[r] = bc[r];
    }
    return bad;
}
template <int D>
__device__ __forceinline__ int embed_pixel(const LatticeFeat &F, int i, int N, const unsigned char *im,
                                           uint32_t (&keys)[D + 1][KeyWords<D>::value], float (&bc_out)[D + 1]) {
    float pr = 0.0f, pg = 0.0f, pb = 0.0f;
    if (D == 5 && i < N) {
        const unsigned char *px = im + (size_t)i * 3;
        pr = (float)px[0]; pg = (float)px[1]; pb = (float)px[2];
    }
    return embed_pixel_rgb<D>(F, i, N, pr, pg, pb, keys, bc_out);
}

}  // namespace dsrg
