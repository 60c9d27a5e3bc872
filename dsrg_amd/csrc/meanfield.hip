// Dense-CRF mean-field inference on gfx950.
//
// Replaces DenseCRF::inference + expAndNormalize (CRF/src/densecrf.cpp:98-131),
// PairwisePotential::apply / DenseKernel::filter (CRF/src/pairwise.cpp:63-80,
// 173-178), PottsCompatibility::apply (CRF/src/labelcompatibility.cpp:46-48) and
// Permutohedral::sseCompute — splat / blur / slice (CRF/src/permutohedral.cpp:
// 529-589).
//
// Design: label planes are independent inside the filter, so one
// workgroup owns CPW label planes of one (image, kernel) lattice and keeps their
// lattice values in LDS for the whole splat -> (d+1) blur passes -> slice chain;
// only the per-pixel messages travel through HBM/L2.  A second, per-pixel kernel
// combines unary + weighted messages and renormalises over the labels; when the
// Gaussian lattice is pixel-local (training scale, see lattice.hip) that kernel
// also forms the Gaussian message itself, in registers, and the filter launch
// carries bilateral workgroups only.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "common.h"

// compile-time experiments on the filter kernel: make EXP=n builds libdsrg_hip.expN.so next to the shipped library and the
// tools select it with DSRG_LIB (A/B on one box).  Measured and adopted in round 3: blur gathers batched per group of 5
// slots (-0.96 us per workgroup), product gathers batched (-0.14); measured and dropped: a workgroup barrier behind the q
// loads (+0.3), the first row term of every slot batched (0.0), term-major row sums (+1.5), neighbour-word descriptors that
// end at M (0.0), LDS-only barriers (0.0).  Adopted later: the extras' index words and the first axis' neighbour words
// issued behind the first barrier (-0.6, -0.2).  profiles/r03_filter_ab.txt
#ifndef DSRG_EXP
#define DSRG_EXP 0
#endif

namespace dsrg {

// ---------------------------------------------------------------------------------
// filter kernel: out_k[b][c][i] = norm_k[i] * (K_k (norm_k . Q[b][c]))[i]
// grid: one block per (label group, lattice); lattices of one image/kernel share an
// XCD (block id % 8) so their index arrays stay in one L2.
enum : int {
    kOptLocalGauss = 1,   // a pixel-local Gaussian lattice is evaluated by the update kernel: its units exit at once
    kOptSlotGuard = 2,    // skip a thread's vertex slots beyond the lattice's actual size (workgroup-uniform test)
    kOptSeq = 4,          // <= 2 label planes: Permutohedral::seqCompute's arithmetic (permutohedral.cpp:476-527 via :600-601) — blur
                          // summed in double, slice as (w * value) * alpha.  Set by plan_filter, one-plane workgroups only
    kOptNormPass = 8,     // the build's normalisation pass (pairwise.cpp:44,54-57): the input is a plane of ones, no norm is
                          // applied, and out = 1/sqrt(K 1 + 1e-20) is the lattice's norm vector.  With kOptSeq, one plane
};
struct FilterArgs {
    LatticeView Lg, Lb;      // Gaussian (shared by all images, nlat==1) and bilateral (per image)
    const float *q;          // (B,C,N)
    float *msg_g, *msg_b;    // (B,C,N)
    int B, C, N;
    int nblk_b;              // bilateral blocks: nblk_xcd + groups_b * (B - lat_stride) when B > lat_stride
    int nblk_xcd;            // the first groups_b * lat_stride blocks: index = group * lat_stride + image
    int lat_stride;          // a multiple of 8, so that the blocks of one image share blockIdx % 8 (one XCD and its L2)
    int groups_b;            // ceil(C / CPW_B)
    int groups_g;            // ceil(C / CPW_G)
    int nunits;              // nblk_b bilateral units (incl. the padding of the XCD map) + groups_g * B Gaussian units
    int lds_bytes;           // dynamic LDS of the launch: value buffer(s) from the start, input planes [N] at the end
    int opts;                // kOpt* bits
    unsigned long long *dbg; // optional per-workgroup phase timestamps (100 MHz wall clock), 2 x 16 per block
};

// What bounds this kernel (measured with per-phase timestamps, tools/filter_trace.py): not HBM and
// not LDS bandwidth alone but exposed memory latency plus the L1's 64 B/clk — a dependent global load
// costs ~1.5 us here and a phase of LDS work only ~0.3 us.  Hence the structure:
//   * every index word a thread will need (splat entries, CSR row bounds, the neighbour words of
//     ALL blur axes, slice corners) is fetched up front / one phase ahead with unconditional SRSRC
//     buffer loads (per-lane offset in one VGPR, strides in the scalar offset, out-of-range reads
//     return 0) — one exposed round trip per lattice instead of one per phase;
//   * label planes are interleaved [vertex][CPW] in LDS and moved 8 bytes at a time (CPW = 2);
//   * one workgroup per CU (LDS); bilateral blocks (2 label planes of one image) first in the grid, Gaussian blocks
//     (4 planes) after them — none at training scale, where the update kernel evaluates the pixel-local lattice.
template <int CPW> struct PlaneVec;
template <> struct PlaneVec<1> { using type = float; };
template <> struct PlaneVec<2> { using type = float2; };
template <> struct PlaneVec<4> { using type = float4; };
__device__ __forceinline__ float pv_get(float v, int) { return v; }
__device__ __forceinline__ float pv_get(float2 v, int c) { return c ? v.y : v.x; }
__device__ __forceinline__ float pv_get(float4 v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }
__device__ __forceinline__ void pv_set(float &v, int, float x) { v = x; }
__device__ __forceinline__ void pv_set(float2 &v, int c, float x) { if (c) v.y = x; else v.x = x; }
__device__ __forceinline__ void pv_set(float4 &v, int c, float x) {
    if (c == 0) v.x = x; else if (c == 1) v.y = x; else if (c == 2) v.z = x; else v.w = x;
}

template <int V> using IC = std::integral_constant<int, V>;

#if DSRG_EXP & 16
#define g_exp_qimg (exp_qimg)
#define g_exp_C (exp_C)
#endif
#define DSRG_STAMP(i_) do { if (dbg && tid == 0) dbg[(i_)] = wall_clock64(); } while (0)

// one lattice (dimension D, index li of set L), planes [c0, c0+nc) of image b:
//   out[c][i] = norm[i] * (K (norm . q[c]))[i]
template <int CPW, int VPT, int PPT, int D, bool SEQ>
__device__ __forceinline__ void filter_lattice(const LatticeView &L, int li, const float *__restrict__ qb,
                                               float *__restrict__ out, int nc, int N,
                                               typename PlaneVec<CPW>::type *val,
                                               typename PlaneVec<CPW>::type *inq, unsigned long long *dbg,
                                               int lds_elems, int opts
#if DSRG_EXP & 16
                                               , const float *exp_qimg, int exp_C
#endif
                                               ) {
    using vec_t = typename PlaneVec<CPW>::type;
    constexpr int D1 = D + 1;
    constexpr bool DEEP = VPT <= 10;             // all index words of a thread fit the register file
    constexpr int KC = DEEP ? VPT : 8;           // vertices per chunk of index loads otherwise
    constexpr int NCH = (VPT + KC - 1) / KC;
    constexpr int RING = 3;
    const int tid = (int)threadIdx.x;
    const int Mcap = L.Mcap;
    const float *norm = L.norm + (size_t)li * N;

    const uint32_t nb_bytes = sizeof(uint32_t) * (uint32_t)Mcap;

#if DSRG_EXP & 8
    const int M = L.M[li];
    const int lat_flags = L.flags[li];
    const int X = L.nextra[li];
    asm volatile("" :: "s"(M), "s"(lat_flags), "s"(X));     // consumed (waited for) here, ahead of the vector loads
#endif
    const rsrc_t r_rs = make_rsrc(L.row_start + (size_t)li * (Mcap + 2), sizeof(uint16_t) * (size_t)(Mcap + 2));
    const rsrc_t r_fp = make_rsrc(L.first_pix + (size_t)li * Mcap, sizeof(uint16_t) * (size_t)Mcap);
    const rsrc_t r_fw = make_rsrc(L.first_w + (size_t)li * Mcap, sizeof(float) * (size_t)Mcap);
    const rsrc_t r_vid = make_rsrc(L.vid + (size_t)li * D1 * N, sizeof(uint16_t) * (size_t)D1 * N);
    const rsrc_t r_bary = make_rsrc(L.bary + (size_t)li * D1 * N, sizeof(float) * (size_t)D1 * N);
    const rsrc_t r_norm = make_rsrc(norm, sizeof(float) * (size_t)N);
    const rsrc_t r_q = make_rsrc(qb, sizeof(float) * (size_t)nc * N);
    const uint32_t *nb_base = L.nb + (size_t)li * D1 * Mcap;
    DSRG_STAMP(0);

    // ---- stage A: everything that does not depend on anything, in one burst
    float qv[PPT][CPW], nrm[PPT];
#pragma unroll
    for (int p = 0; p < PPT; p++) {
        nrm[p] = ld_f32(r_norm, (uint32_t)tid * 4u, (uint32_t)p * (kWG * 4u));
#pragma unroll
        for (int c = 0; c < CPW; c++)
            qv[p][c] = ld_f32(r_q, (uint32_t)tid * 4u, (uint32_t)p * (kWG * 4u) + (uint32_t)c * (uint32_t)N * 4u);
    }
#if DSRG_EXP & 16
    // PROTOTYPE, measurement only (round-4 review item 3: "fold mf_update_split_kernel into the head of the next filter launch"):
    // the least a folded head can do — per pixel ONE pre-combined logit plane per label (21 loads where the update kernel has
    // 63), the fp64-rounded exp of every label and their label-order sum (the softmax denominator every plane pair of the image
    // needs).  The result only feeds an empty asm, the separate update kernel still runs: this build times the head's cost
    // inside the filter workgroup's critical path, nothing else (profiles/r05_filter_ab.txt).
    {
        const float *qimg = g_exp_qimg;
        const int Call = g_exp_C;
        const rsrc_t r_all = make_rsrc(qimg, sizeof(float) * (size_t)Call * N);
        float keep = 0.0f;
#pragma unroll
        for (int p = 0; p < PPT; p++) {
            float mx = -INFINITY, sum = 0.0f;
            for (int c = 0; c < Call; c++) mx = fmaxf(mx, ld_f32(r_all, (uint32_t)tid * 4u, ((uint32_t)p * kWG + (uint32_t)c * (uint32_t)N) * 4u));
            for (int c = 0; c < Call; c++)
                sum = sum + exp_cr(ld_f32(r_all, (uint32_t)tid * 4u, ((uint32_t)p * kWG + (uint32_t)c * (uint32_t)N) * 4u) - mx);
            keep += sum;
        }
        asm volatile("" :: "v"(keep));
    }
#endif
    const bool norm_pass = SEQ && (opts & kOptNormPass);
    if (norm_pass) {                          // (the loads above hit valid memory — q aliases the norm vector — and are dropped)
#pragma unroll
        for (int p = 0; p < PPT; p++) {
            nrm[p] = 1.0f;
#pragma unroll
            for (int c = 0; c < CPW; c++) qv[p][c] = 1.0f;
        }
    }
    // splat, vertex-major: the first contributor (pixel, weight) of my vertices v_k = tid + k*1024 and their CSR rows
    uint32_t fpx[KC], rs0[KC], rs1[KC];
    float fw[KC];
    auto load_rows = [&](int ch) {
#pragma unroll
        for (int k = 0; k < KC; k++) {
            const uint32_t sk = (uint32_t)(ch * KC + k);
            fpx[k] = ld_u16(r_fp, (uint32_t)tid * 2u, sk * (kWG * 2u));
            fw[k] = ld_f32(r_fw, (uint32_t)tid * 4u, sk * (kWG * 4u));
            rs0[k] = ld_u16(r_rs, (uint32_t)tid * 2u, sk * (kWG * 2u));
        }
        // row end = the next vertex's row start: the neighbouring lane holds it; only the last lane of a wave loads it
        if ((tid & 63) == 63) {
#pragma unroll
            for (int k = 0; k < KC; k++) rs1[k] = ld_u16(r_rs, (uint32_t)tid * 2u, (uint32_t)(ch * KC + k) * (kWG * 2u) + 2u);
        }
    };
    auto finish_rows = [&]() {
#pragma unroll
        for (int k = 0; k < KC; k++) {
            const uint32_t up = __shfl_down(rs0[k], 1, 64);
            if ((tid & 63) != 63) rs1[k] = up;
        }
    };
    load_rows(0);
    // neighbour words n1 | n2<<16 of my vertices: a ring of RING axes, fetched RING-1 passes ahead of
    // their use (a blur pass is shorter than one memory round trip)
    uint32_t nbw[DEEP ? RING : 1][KC];
    auto load_axis = [&](int j) {
        const rsrc_t r_nb = make_rsrc(nb_base + (size_t)j * Mcap, nb_bytes);
#pragma unroll
        for (int k = 0; k < KC; k++)
            nbw[j % RING][k] = ld_u32(r_nb, (uint32_t)tid * 4u, (uint32_t)k * (kWG * 4u));
    };
    finish_rows();

    // the lattice's size, flags and extras count are consumed only here, behind the burst: a scalar round trip in front of
    // the first vector load would add its latency to every workgroup's start
#if !(DSRG_EXP & 8)
    const int M = L.M[li];
    const int lat_flags = L.flags[li];
    const int X = L.nextra[li];              // entries beyond the first of their row
#endif
    // slots k with k * kWG >= Mlim hold no vertex of this lattice: skipped when the guard is on
    const int Mlim = (opts & kOptSlotGuard) ? M : (VPT * kWG);
    constexpr bool seq = SEQ;                 // kOptSeq arithmetic: its own instantiation (a runtime test cost the
                                              // one-plane workgroups of a lone image 1 us per launch)
    // the extras (entry-parallel: x_k = tid + k*1024 < X), through descriptors that end at X: slots beyond read 0 for free
    const rsrc_t r_xp = make_rsrc(L.x_pix + (size_t)li * D1 * N, sizeof(uint16_t) * (size_t)X);
    const rsrc_t r_xw = make_rsrc(L.x_w + (size_t)li * D1 * N, sizeof(float) * (size_t)X);
    uint32_t xpx[KC];
    float xw[KC];
    auto load_extras = [&](int ch) {
#pragma unroll
        for (int k = 0; k < KC; k++) {
            if ((ch * KC + k) * kWG >= X) { xpx[k] = 0u; xw[k] = 0.0f; continue; }   // (workgroup-uniform)
            xpx[k] = ld_u16(r_xp, (uint32_t)tid * 2u, (uint32_t)(ch * KC + k) * (kWG * 2u));
            xw[k] = ld_f32(r_xw, (uint32_t)tid * 4u, (uint32_t)(ch * KC + k) * (kWG * 4u));
        }
    };

    if (lat_flags & 1) {
        // Diagonal lattice: every simplex corner is private to its pixel and has no blur neighbour, so splat, blur and
        // slice collapse to per-pixel arithmetic — evaluated here in the general path's operation
        // order (products, 0 + p, val + 0.5*(0+0), ordered slice sum), hence bit-identical to it.
        const float alpha = 1.0f / (1.0f + exp2f(-(float)D));
        float bws[PPT][D1];
#pragma unroll
        for (int p = 0; p < PPT; p++)
#pragma unroll
            for (int r = 0; r < D1; r++)
                bws[p][r] = ld_f32(r_bary, (uint32_t)tid * 4u, ((uint32_t)p * kWG + (uint32_t)r * (uint32_t)N) * 4u);
#pragma unroll
        for (int p = 0; p < PPT; p++) {
            const int i = tid + p * kWG;
            if (i < N) {
#pragma unroll
                for (int c = 0; c < CPW; c++) {
                    if (c < nc) {
                        const float x = qv[p][c] * nrm[p];
                        float acc = 0.0f;
#pragma unroll
                        for (int r = 0; r < D1; r++) {
                            float v = 0.0f + bws[p][r] * x;      // splat into an empty vertex
                            v = v + 0.5f * (0.0f + 0.0f);        // d+1 blur passes without neighbours
                            if (SEQ) { float t = bws[p][r] * v; t = t * alpha; acc = acc + t; }
                            else acc = acc + (bws[p][r] * alpha) * v; // slice
                        }
                        out[(size_t)c * N + i] = norm_pass ? (float)(1.0 / sqrt((double)acc + 1e-20)) : acc * nrm[p];
                    }
                }
            }
        }
        DSRG_STAMP(11);
        if (dbg && tid == 0) dbg[12] = (unsigned long long)M;
        return;
    }

    // in = Q * norm   (pairwise.cpp:66)
#pragma unroll
    for (int p = 0; p < PPT; p++) {
        const int i = tid + p * kWG;
        if (i < N) {
            vec_t x;
#pragma unroll
            for (int c = 0; c < CPW; c++) pv_set(x, c, (c < nc) ? qv[p][c] * nrm[p] : 0.0f);
            inq[i] = x;
        }
    }
    load_extras(0);                          // behind the input planes: nothing the first barrier waits for queues behind them
    __syncthreads();
    DSRG_STAMP(1);
    // the first axis' neighbour words are not needed before the first blur pass: issued behind the input planes' barrier,
    // they no longer queue in the L1 ahead of other waves' q loads (-0.44 us on the way to that barrier)
    if constexpr (DEEP) load_axis(0);

    // ---- splat (permutohedral.cpp:545-553) in the reference's accumulation order without a dependent global load and
    // without float atomics.  Vertex v's value is the ordered sum of its row of (pixel, weight) entries (sorted by the
    // reference's visiting order at build time).  Most rows hold ONE entry: that first term is formed vertex-parallel,
    // straight from the input planes (0 + w * in[pixel]); only the further entries ("extras", X of them) go through an
    // entry-parallel products pass into LDS and are then added row by row, in order.  `prod` aliases the first value buffer.
    vec_t *prod = val;
    vec_t sacc[VPT];
    // LDS gathers go out in batches of slots with ONE workgroup-uniform guard per batch (a guard per slot serialises the
    // gathers: +0.96 us per workgroup).  Batches of 5, 4 and 1 slots: a 41x41 lattice has 10 slots of 1024 vertices and
    // typically 7 800 - 8 900 of them, so the last slot is empty and the one before partly
    constexpr int GS = 5, kG1 = KC < 5 ? KC : 5, kG2 = KC < 9 ? KC : 9;
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
        if (ch > 0) load_extras(ch);
#pragma unroll
        for (int k0 = 0; k0 < KC; k0 += GS) {
            if ((ch * KC + k0) * kWG < X) {
                vec_t xin[GS];
#pragma unroll
                for (int g = 0; g < GS; g++)
                    if (k0 + g < KC) xin[g] = inq[min((int)xpx[k0 + g], N - 1)];           // gathers in flight together
#pragma unroll
                for (int g = 0; g < GS; g++) {
                    if (k0 + g < KC) {
                        const int x = tid + (ch * KC + k0 + g) * kWG;
                        vec_t p;
#pragma unroll
                        for (int c = 0; c < CPW; c++) pv_set(p, c, xw[k0 + g] * pv_get(xin[g], c));
                        if (x < X) prod[x] = p;
                    }
                }
            }
        }
    }
    // first terms (they read the input planes, like the products above: same phase)
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
        if (ch > 0) { load_rows(ch); finish_rows(); }
        auto first_group = [&](auto k0c, auto k1c) {
            constexpr int k0 = decltype(k0c)::value, G = decltype(k1c)::value - k0;
            if constexpr (G > 0) {
                if (ch * KC + k0 < VPT && (ch * KC + k0) * kWG < Mlim) {
                    vec_t xin[G];
#pragma unroll
                    for (int g = 0; g < G; g++) xin[g] = inq[min((int)fpx[k0 + g], N - 1)];
#pragma unroll
                    for (int g = 0; g < G; g++) {
                        if (ch * KC + k0 + g < VPT) {
#pragma unroll
                            for (int c = 0; c < CPW; c++)
                                pv_set(sacc[ch * KC + k0 + g], c, 0.0f + fw[k0 + g] * pv_get(xin[g], c));
                        }
                    }
                }
            }
        };
        first_group(IC<0>{}, IC<kG1>{});
        first_group(IC<kG1>{}, IC<kG2>{});
        first_group(IC<kG2>{}, IC<KC>{});
    }
    if constexpr (DEEP) { if (1 < D1) load_axis(1); }
    DSRG_STAMP(2);
    if (X > 0) {                                         // (workgroup-uniform) some row has more than one entry
        __syncthreads();                                 // the products are in place
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            if (NCH > 1) { load_rows(ch); finish_rows(); }
            // extras of row v: [rs0 - v, rs1 - v - 1) — every row before a non-empty row is non-empty
            uint32_t t0[KC], t1[KC];
#pragma unroll
            for (int k = 0; k < KC; k++) {
                const int v = tid + (ch * KC + k) * kWG;
                t0[k] = rs0[k] - (uint32_t)v;
                t1[k] = (v < M && rs1[k] > rs0[k]) ? rs1[k] - (uint32_t)v - 1u : t0[k];
            }
#pragma unroll
            for (int k = 0; k < KC; k++) {
                if (ch * KC + k < VPT && (ch * KC + k) * kWG < Mlim) {
                    float s[CPW];
#pragma unroll
                    for (int c = 0; c < CPW; c++) s[c] = pv_get(sacc[ch * KC + k], c);
                    for (uint32_t t = t0[k]; t < t1[k]; t++) {
                        const vec_t p = prod[t];
#pragma unroll
                        for (int c = 0; c < CPW; c++) s[c] = s[c] + pv_get(p, c);
                    }
#pragma unroll
                    for (int c = 0; c < CPW; c++) pv_set(sacc[ch * KC + k], c, s[c]);
                }
            }
        }
    }
    DSRG_STAMP(3);
    if constexpr (DEEP) { if (2 < D1) load_axis(2); }
    // Two value buffers when the region holds them (the input planes are dead after this phase): an axis then gathers from
    // one and writes the other — one barrier per axis instead of two.  The first values go to the SECOND buffer when the
    // products fit inside the first: nobody reads that memory in this phase, so no barrier is needed in front of the writes.
    const int vstride = (M + 2) & ~1;
    const bool pingpong = 2 * vstride <= lds_elems;
    const bool direct = pingpong && X <= vstride && 2 * vstride + N <= lds_elems;   // (the input planes end the region)
    vec_t *cur = direct ? val + vstride : val, *nxt = pingpong ? (direct ? val : val + vstride) : val;
    if (!direct) __syncthreads();                        // every row of products (and every input value) has been consumed
#pragma unroll
    for (int k = 0; k < VPT; k++) {
        const int v = tid + k * kWG;
        if (k * kWG < Mlim && v < M) cur[v] = sacc[k];
    }
    {                                                   // zero sentinel = "no neighbour" (permutohedral.cpp:561-562): slot M
        vec_t z;
#pragma unroll
        for (int c = 0; c < CPW; c++) pv_set(z, c, 0.0f);
        if (tid == 0) { cur[M] = z; if (!direct) nxt[M] = z; }
    }
    __syncthreads();
    if (direct && tid == 0) {                           // the first buffer's sentinel: the products there are dead now
        vec_t z;
#pragma unroll
        for (int c = 0; c < CPW; c++) pv_set(z, c, 0.0f);
        nxt[M] = z;
    }
    DSRG_STAMP(4);

    // ---- blur along the d+1 lattice axes (permutohedral.cpp:556-569): Jacobi per axis — new values
    // held in registers between the read barrier and the write barrier
    constexpr int kSliceAxisA = D1 >= 3 ? D1 - 3 : 0, kSliceAxisB = D1 >= 2 ? D1 - 2 : 0, kSliceHalf = (D1 + 1) / 2;
    static_assert(kSliceAxisA != kSliceAxisB, "the two halves of the slice corners need two distinct axes");
    uint32_t sv[PPT][D1];
    float sw[PPT][D1];
#pragma unroll
    for (int j = 0; j < D1; j++) {
        // sacc[k] holds the current value of my vertex v_k (no LDS read for it)
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            if constexpr (!DEEP) {
                const rsrc_t r_nb = make_rsrc(nb_base + (size_t)j * Mcap, nb_bytes);
#pragma unroll
                for (int k = 0; k < KC; k++)
                    nbw[0][k] = ld_u32(r_nb, (uint32_t)tid * 4u, (uint32_t)(ch * KC + k) * (kWG * 4u));
            }
            auto blur_group = [&](auto k0c, auto k1c) {
                constexpr int k0 = decltype(k0c)::value, G = decltype(k1c)::value - k0;
                if constexpr (G > 0) {
                    if (ch * KC + k0 < VPT && (ch * KC + k0) * kWG < Mlim) {
                        vec_t x1[G], x2[G];
#pragma unroll
                        for (int g = 0; g < G; g++) {
                            const uint32_t word = nbw[DEEP ? j % RING : 0][k0 + g];
                            x1[g] = cur[(int)(word & 0xffffu)];
                            x2[g] = cur[(int)(word >> 16)];
                        }
#pragma unroll
                        for (int g = 0; g < G; g++) {
                            if (ch * KC + k0 + g < VPT) {
#pragma unroll
                                for (int c = 0; c < CPW; c++) {
                                    float s = pv_get(x1[g], c) + pv_get(x2[g], c);
                                    if (seq) {       // new = old + 0.5 * (n1 + n2) with a double literal
                                        pv_set(sacc[ch * KC + k0 + g], c,
                                               (float)((double)pv_get(sacc[ch * KC + k0 + g], c) + 0.5 * (double)s));
                                    } else {
                                        s = 0.5f * s;
                                        pv_set(sacc[ch * KC + k0 + g], c, pv_get(sacc[ch * KC + k0 + g], c) + s);
                                    }
                                }
                            }
                        }
                    }
                }
            };
            blur_group(IC<0>{}, IC<kG1>{});
            blur_group(IC<kG1>{}, IC<kG2>{});
            blur_group(IC<kG2>{}, IC<KC>{});
        }
        if constexpr (DEEP) { if (j + RING < D1) load_axis(j + RING); }     // this axis' ring slot is free
        // slice corners, one and two passes ahead of their use and after the last ring fetch: half of them per axis (all 24
        // loads of a thread on one axis held the waves' LDS work up by 0.7 us)
        if (j == kSliceAxisA || j == kSliceAxisB) {
            const int r0 = (j == kSliceAxisA) ? 0 : kSliceHalf, r1 = (j == kSliceAxisA) ? kSliceHalf : D1;
#pragma unroll
            for (int p = 0; p < PPT; p++) {
#pragma unroll
                for (int r = 0; r < D1; r++) {
                    if (r >= r0 && r < r1) {
                        const uint32_t so = (uint32_t)p * kWG + (uint32_t)r * (uint32_t)N;
                        sv[p][r] = ld_u16(r_vid, (uint32_t)tid * 2u, so * 2u);
                        sw[p][r] = ld_f32(r_bary, (uint32_t)tid * 4u, so * 4u);
                    }
                }
            }
        }
        if (pingpong) {
#pragma unroll
            for (int k = 0; k < VPT; k++) {
                const int v = tid + k * kWG;
                if (k * kWG < Mlim && v < M) nxt[v] = sacc[k];
            }
            __syncthreads();                         // gathers of this axis done, values of the next one in place
            vec_t *t = cur; cur = nxt; nxt = t;
        } else {
            __syncthreads();                         // all gathers of this axis are done
#pragma unroll
            for (int k = 0; k < VPT; k++) {
                const int v = tid + k * kWG;
                if (k * kWG < Mlim && v < M) cur[v] = sacc[k];
            }
            __syncthreads();
        }
        DSRG_STAMP(5 + j);
    }

    // ---- slice (permutohedral.cpp:571-584), then out * norm (pairwise.cpp:79)
    const float alpha = 1.0f / (1.0f + exp2f(-(float)D));
#pragma unroll
    for (int p = 0; p < PPT; p++) {
        const int i = tid + p * kWG;
        if (i < N) {
            float acc[CPW];
#pragma unroll
            for (int c = 0; c < CPW; c++) acc[c] = 0.0f;
#pragma unroll
            for (int r = 0; r < D1; r++) {
                const vec_t x = cur[min(sv[p][r], (uint32_t)M)];
                if (seq) {
#pragma unroll
                    for (int c = 0; c < CPW; c++) { float t = sw[p][r] * pv_get(x, c); t = t * alpha; acc[c] = acc[c] + t; }
                } else {
                    const float w = sw[p][r] * alpha;
#pragma unroll
                    for (int c = 0; c < CPW; c++) acc[c] = acc[c] + w * pv_get(x, c);
                }
            }
#pragma unroll
            for (int c = 0; c < CPW; c++)
                if (c < nc) out[(size_t)c * N + i] = norm_pass ? (float)(1.0 / sqrt((double)acc[c] + 1e-20)) : acc[c] * nrm[p];
        }
    }
    DSRG_STAMP(11);
    if (dbg && tid == 0) dbg[12] = (unsigned long long)M;
}

// One launch filters every label plane of every image through both lattices.  Work units = workgroups: first the
// bilateral ones (CPW_B planes of one image: 6 axes, M ~ 2-6 N vertices, the long ones), then the Gaussian ones (CPW_G planes
// of one image through the lattice all images share: 3 axes), handed out by the hardware dispatcher (a software unit queue
// was measured and lost, profiles/r02_filter_queue_ab.txt).
// (DSRG_EXP & 64, measured and not adopted — profiles/r06_filter_ab.txt: the kernel built for TWO workgroups per CU — 64 VGPRs
// at 1 024 threads, 112 bytes of scratch per lane for the 10-vertex instantiation — so that the 336 one-plane workgroups of a
// 16-image batch are resident at once instead of 176 plane pairs on 176 CUs)
#if DSRG_EXP & 64
#define DSRG_FILTER_BOUNDS __launch_bounds__(kWG, 8)
#else
#define DSRG_FILTER_BOUNDS __launch_bounds__(kWG)
#endif
template <int CPW_B, int CPW_G, int VPT_B, int PPT, bool SEQ>
__global__ DSRG_FILTER_BOUNDS void mf_filter_kernel(FilterArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int VPT_G = (VPT_B + 1) / 2;                 // Mcap_gauss = Mcap_bilateral / 2
    unsigned long long *dbg = a.dbg ? a.dbg + (size_t)blockIdx.x * 32 : nullptr;
    const int unit = blockIdx.x;
    if (unit < a.nblk_b) {
        using vec_t = typename PlaneVec<CPW_B>::type;
        // units of one image share unit % 8, i.e. one XCD and its L2; the images beyond the last multiple of 8 (B = 20:
        // images 16..19) are laid out image-major so that they spread evenly over the XCDs
        int g, b;
        if (unit < a.nblk_xcd) { g = unit / a.lat_stride; b = unit % a.lat_stride; }
        else { const int rel = unit - a.nblk_xcd; b = a.lat_stride + rel / a.groups_b; g = rel % a.groups_b; }
        if (b >= a.B) return;
        const int c0 = g * CPW_B, nc = min(CPW_B, a.C - c0);
        vec_t *val = reinterpret_cast<vec_t *>(smem);                      // value buffer(s), label-interleaved
        vec_t *inq = reinterpret_cast<vec_t *>(smem + a.lds_bytes) - a.N;  // [N] at the end of the region
        const size_t o = ((size_t)b * a.C + c0) * a.N;
        filter_lattice<CPW_B, VPT_B, PPT, 5, SEQ>(a.Lb, b, a.q + o, a.msg_b + o, nc, a.N, val, inq, dbg,
                                             a.lds_bytes / (int)sizeof(vec_t), a.opts
#if DSRG_EXP & 16
                                             , a.q + (size_t)b * a.C * a.N, SEQ ? 1 : a.C
#endif
                                             );
    } else {
        using vec_t = typename PlaneVec<CPW_G>::type;
        if ((a.opts & kOptLocalGauss) && (a.Lg.flags[0] & kLatticeLocal)) return;   // the update kernel forms this message
        const int rel = unit - a.nblk_b;
        const int b = rel / a.groups_g, g = rel % a.groups_g;
        const int c0 = g * CPW_G, nc = min(CPW_G, a.C - c0);
        vec_t *val = reinterpret_cast<vec_t *>(smem);
        vec_t *inq = reinterpret_cast<vec_t *>(smem + a.lds_bytes) - a.N;
        const size_t o = ((size_t)b * a.C + c0) * a.N;
        filter_lattice<CPW_G, VPT_G, PPT, 2, SEQ>(a.Lg, 0, a.q + o, a.msg_g + o, nc, a.N, val, inq, dbg ? dbg + 16 : nullptr,
                                             a.lds_bytes / (int)sizeof(vec_t), a.opts
#if DSRG_EXP & 16
                                             , a.q + (size_t)b * a.C * a.N, 0
#endif
                                             );
    }
}
#undef DSRG_STAMP

// ---------------------------------------------------------------------------------
// Pixel-local Gaussian lattice (flag kLatticeLocal, lattice.hip: lattice_local_kernel).  At training scale the spatial
// kernel has sigma = 0.25 px: the three corners of a pixel's simplex are private to the pixel (one contributor each) and
// their only blur neighbours are each other — along every axis exactly one pair of the three exchanges half its values.
// The build relabels the corners of each pixel so that the pairs are (0,1), (1,2), (2,0) for axes 0, 1, 2, stores the
// weights in that order and the relabelled index z of original corner 2 (the LAST term of the slice sum; float addition
// commutes, so only the last term's identity matters for the rounding).  What follows is DenseKernel::filter
// (pairwise.cpp:63-80) through sseCompute (permutohedral.cpp:544-585) for one pixel and one label in the general path's
// operation order with the exact zeros left out (0 + x, x + 0.5*(0 + 0)): bit-identical to filter_lattice's result.
struct GaussLocal {
    float nv, b0, b1, b2;   // norm, relabelled barycentric weights
    float a0, a1, a2;       // b * alpha (the slice weights)
    bool z0, z2;            // original corner 2 is relabelled corner 0 / 2 (else 1)
};
__device__ __forceinline__ GaussLocal gauss_local_load(const float4 *__restrict__ la, const uint32_t *__restrict__ lz, int i) {
    const float4 w = la[i];
    const uint32_t z = lz[i];
    const float alpha = 1.0f / (1.0f + exp2f(-2.0f));
    GaussLocal g;
    g.nv = w.x; g.b0 = w.y; g.b1 = w.z; g.b2 = w.w;
    g.a0 = w.y * alpha; g.a1 = w.z * alpha; g.a2 = w.w * alpha;
    g.z0 = z == 0u; g.z2 = z == 2u;
    return g;
}
__device__ __forceinline__ float gauss_local_apply(const GaussLocal &g, float q) {
    const float x = q * g.nv;                          // pairwise.cpp:66
    float u0 = g.b0 * x, u1 = g.b1 * x, u2 = g.b2 * x; // splat into private vertices
    float n0, n1, n2;
    n0 = u0 + 0.5f * u1; n1 = u1 + 0.5f * u0; u0 = n0; u1 = n1;       // axis 0: pair (0,1)
    n1 = u1 + 0.5f * u2; n2 = u2 + 0.5f * u1; u1 = n1; u2 = n2;       // axis 1: pair (1,2)
    n2 = u2 + 0.5f * u0; n0 = u0 + 0.5f * u2; u2 = n2; u0 = n0;       // axis 2: pair (2,0)
    const float t0 = g.a0 * u0, t1 = g.a1 * u1, t2 = g.a2 * u2;      // slice terms
    const float A = g.z0 ? t1 : t0, Bv = g.z2 ? t1 : t2, Z = g.z0 ? t0 : (g.z2 ? t2 : t1);
    float acc = A + Bv;
    acc = acc + Z;
    return acc * g.nv;                                 // pairwise.cpp:79
}

// the Gaussian message of a pixel-local lattice on its own (dsrg_ctx_filter_once; the inference loop never launches it)
__global__ __launch_bounds__(256) void gauss_local_kernel(const float4 *__restrict__ la, const uint32_t *__restrict__ lz,
                                                          const float *__restrict__ q, float *__restrict__ out, int planes, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const GaussLocal g = gauss_local_load(la, lz, i);
    for (int c = blockIdx.y; c < planes; c += gridDim.y) out[(size_t)c * N + i] = gauss_local_apply(g, q[(size_t)c * N + i]);
}

// ---------------------------------------------------------------------------------
// per-pixel update: Q = expAndNormalize( -U - sum_k (-w_k msg_k) )   (densecrf.cpp:98-106,122-128)
// The last iteration additionally emits the layer outputs of pylayers.py:84-88.
// kUpdParts threads per pixel: lanes = 64 consecutive pixels (coalesced plane loads), wave p of the workgroup takes the
// labels c = p, p + kUpdParts, ...  (a one-thread-per-pixel kernel ran a ~1.4 us chain of 21 dependent fp64 exps per thread
// at less than one wave per SIMD); the column max, the label-order sum and numpy's pairwise sum go through LDS in exactly
// the reference's order.  msg_g_or_q: the normalised Gaussian message, or — `loc` given and the lattice flagged
// pixel-local — the marginals the bilateral filter just read, from which the Gaussian message is formed here.
struct UpdLocal {
    const float4 *la;          // [N] (norm, relabelled weights) of the shared Gaussian lattice
    const uint32_t *lz;        // [N]
    const int *flags;          // the Gaussian lattice's flag word; null = never local
};
constexpr int kUpdParts = 8, kUpdPix = 64;      // (4 parts: +3.5 us per step at 16 images, +7.5 for a lone image; 16: +4.5 / +2)
template <int CT, bool USE_MSGS>   // CT = compile-time bound on C
__global__ __launch_bounds__(kUpdParts * kUpdPix) void mf_update_split_kernel(
    const float *__restrict__ neg_unary, const float *__restrict__ msg_g, const float *q_in,
    const float *__restrict__ msg_b, float wg, float wb, float *q_out, double *__restrict__ refined_out,
    float *__restrict__ logq_out, int B, int C, int N, UpdLocal loc) {
    __shared__ float ev[CT][kUpdPix];                      // e, then q per (label, pixel)
    __shared__ float pm[kUpdParts][kUpdPix];               // partial column maxima
    const int px = threadIdx.x & (kUpdPix - 1), part = threadIdx.x >> 6;
    // XCD-affine block map (blockIdx % 8 = the XCD, as in the filter launch): the tiles of image b run on XCD b % 8, the one
    // whose L2 holds the messages the filter workgroups of that image just wrote and from which they will read the marginals
    const int ntile = (N + kUpdPix - 1) / kUpdPix;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int b_map = xcd + 8 * (slot / ntile), i_map = (slot % ntile) * kUpdPix + px;
    const bool live = b_map < B && i_map < N;
    const int b = live ? b_map : 0, i = live ? i_map : 0;
    const size_t base = (size_t)b * C * N + i;
    constexpr int LPT = (CT + kUpdParts - 1) / kUpdParts;
    const bool local = USE_MSGS && loc.flags && (*loc.flags & kLatticeLocal);      // workgroup-uniform
    const float *gsrc = local ? q_in : msg_g;      // q_in may be q_out: a thread reads exactly the elements it writes later
    float t[LPT], mg[LPT], mb[LPT];
#pragma unroll
    for (int k = 0; k < LPT; k++) {                        // all loads first, unconditionally (label index clamped)
        const size_t o = base + (size_t)min(part + k * kUpdParts, C - 1) * N;
        t[k] = neg_unary[o];
        if (USE_MSGS) { mg[k] = gsrc[o]; mb[k] = msg_b[o]; }
    }
    if (local) {
        const GaussLocal g = gauss_local_load(loc.la, loc.lz, i);
#pragma unroll
        for (int k = 0; k < LPT; k++) mg[k] = gauss_local_apply(g, mg[k]);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < LPT; k++) {
        const int c = part + k * kUpdParts;
        float v = t[k];
        if (USE_MSGS) {
            // tmp2 = -w * filter(Q); tmp1 -= tmp2  — Gaussian first, then bilateral
            const float m1 = (-wg) * mg[k];
            v = v - m1;
            const float m2 = (-wb) * mb[k];
            v = v - m2;
        }
        t[k] = (c < C) ? v : -INFINITY;
        mx = fmaxf(mx, t[k]);
    }
    pm[part][px] = mx;
    __syncthreads();
    mx = pm[0][px];
#pragma unroll
    for (int q = 1; q < kUpdParts; q++) mx = fmaxf(mx, pm[q][px]);           // fmax is order-independent
#pragma unroll
    for (int k = 0; k < LPT; k++) {
        const int c = part + k * kUpdParts;
        if (c < C) ev[c][px] = exp_cr(t[k] - mx);
    }
    __syncthreads();
    float sum = 0.0f;                                      // label-order sum, as the reference's column sum
    for (int c = 0; c < C; c++) sum = sum + ev[c][px];
    float q[LPT];
#pragma unroll
    for (int k = 0; k < LPT; k++) {
        const int c = part + k * kUpdParts;
        q[k] = (c < C) ? ev[c][px] / sum : 0.0f;
        if (c < C && live && q_out) q_out[base + (size_t)c * N] = q[k];
    }
    if (refined_out) {
        // pylayers.py:84-88: float64, clip at min_prob, divide by numpy's pairwise label sum, log
        __syncthreads();
#pragma unroll
        for (int k = 0; k < LPT; k++) {
            const int c = part + k * kUpdParts;
            if (c < C) ev[c][px] = q[k];
        }
        __syncthreads();
        auto col = [&](int k) { const double v = (double)ev[k][px]; return v < 0.0001 ? 0.0001 : v; };
        // numpy's float64 add-reduce over a contiguous run of C < 128 values: 8 partial sums, then the tail
        double s;
        if (C < 8) {
            s = 0.0;
            for (int k = 0; k < C; k++) s += col(k);
        } else {
            const int full = C - (C % 8);
            double rs[8];
#pragma unroll
            for (int k = 0; k < 8; k++) rs[k] = col(k);
            for (int k0 = 8; k0 < full; k0 += 8) {
#pragma unroll
                for (int k = 0; k < 8; k++) rs[k] += col(k0 + k);
            }
            s = ((rs[0] + rs[1]) + (rs[2] + rs[3])) + ((rs[4] + rs[5]) + (rs[6] + rs[7]));
            for (int k = full; k < C; k++) s += col(k);
        }
#pragma unroll
        for (int k = 0; k < LPT; k++) {
            const int c = part + k * kUpdParts;
            if (c < C && live) {
                const double qd = (double)q[k];
                const double r = (qd < 0.0001 ? 0.0001 : qd) / s;
                refined_out[base + (size_t)c * N] = r;
                if (logq_out) logq_out[base + (size_t)c * N] = (float)log(r);
            }
        }
    }
}

// ---------------------------------------------------------------------------------
void *g_filter_dbg = nullptr;   // set through dsrg_debug_set_filter_trace (tools only)
// tests / tools: kOpt* bits of the filter launch; -1 = from DSRG_FILTER_OPTS at first use (default: all on)
std::atomic<int> g_filter_opts{-1};
static int filter_opts() {
    int v = g_filter_opts.load(std::memory_order_relaxed);
    if (v < 0) {                                   // any thread may resolve the default: they all compute the same value
        const char *e = getenv("DSRG_FILTER_OPTS");
        v = e ? (atoi(e) & 3) : (kOptLocalGauss | kOptSlotGuard);
        g_filter_opts.store(v, std::memory_order_relaxed);
    }
    return v;
}

template <int CPW_B, int CPW_G, int VPT_B, int PPT, bool SEQ>
static int launch_filter(const FilterArgs &a, int nblocks, size_t lds, hipStream_t stream, Profiler *prof) {
    static LdsGrant granted;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&mf_filter_kernel<CPW_B, CPW_G, VPT_B, PPT, SEQ>), lds, granted);
    if (rc) return rc;
    const bool timed = prof && prof->active && prof->used < prof->cap;
    if (timed) DSRG_HIP_CHECK(hipEventRecord(prof->start[prof->used], stream));
    hipLaunchKernelGGL((mf_filter_kernel<CPW_B, CPW_G, VPT_B, PPT, SEQ>), dim3(nblocks), dim3(kWG), lds, stream, a);
    DSRG_LAUNCH_CHECK();
    if (timed) { DSRG_HIP_CHECK(hipEventRecord(prof->stop[prof->used], stream)); prof->used++; }
    return DSRG_OK;
}

template <int CPW_B, int CPW_G, bool SEQ = false>
static int dispatch_vpt(const FilterArgs &a, int nblocks, size_t lds, int vpt, hipStream_t stream, Profiler *prof) {
    if (vpt <= 4) return launch_filter<CPW_B, CPW_G, 4, 1, SEQ>(a, nblocks, lds, stream, prof);
    if (vpt <= 10) return launch_filter<CPW_B, CPW_G, 10, 2, SEQ>(a, nblocks, lds, stream, prof);
    if (vpt <= 16) return launch_filter<CPW_B, CPW_G, 16, 3, SEQ>(a, nblocks, lds, stream, prof);
    if constexpr (CPW_B > 1) return set_error(DSRG_ERR_UNSUPPORTED, "plane pairs beyond 16 vertices per thread (vpt=%d): plan_filter picks single planes there", vpt);
    if (vpt <= 25) return launch_filter<CPW_B, CPW_G, 25, 5, SEQ>(a, nblocks, lds, stream, prof);      // 65 x 65 (513 / 8 + 1): no scratch
    if (vpt <= 29 && a.N <= 5 * kWG) return launch_filter<CPW_B, CPW_G, 29, 5, SEQ>(a, nblocks, lds, stream, prof);      // (66 .. 70 pixels a side)
    if (vpt <= 32) return launch_filter<CPW_B, CPW_G, 32, 6, SEQ>(a, nblocks, lds, stream, prof);
    return set_error(DSRG_ERR_UNSUPPORTED, "lattice too large for the LDS-resident filter (vpt=%d)", vpt);
}

static int launch_update(const float *neg_unary, const MeanfieldBufs &buf, const UpdLocal &loc, float wg, float wb,
                         int use_msgs, float *q_out, double *refined, float *logq, int B, int C, int N, hipStream_t stream) {
    const int blocks = 8 * ((N + kUpdPix - 1) / kUpdPix) * ((B + 7) / 8);
#define DSRG_UPDS(CT_, UM_)                                                                                               \
    hipLaunchKernelGGL((mf_update_split_kernel<CT_, UM_>), dim3(blocks), dim3(kUpdParts * kUpdPix), 0, stream, neg_unary,     \
                       buf.msg_g, buf.q, buf.msg_b, wg, wb, q_out, refined, logq, B, C, N, loc)
    if (C <= 21) { if (use_msgs) DSRG_UPDS(21, true); else DSRG_UPDS(21, false); }
    else if (C <= 64) { if (use_msgs) DSRG_UPDS(64, true); else DSRG_UPDS(64, false); }
    else { if (use_msgs) DSRG_UPDS(kMaxLabels, true); else DSRG_UPDS(kMaxLabels, false); }
#undef DSRG_UPDS
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

// geometry of one filter launch over B images
struct FilterPlan {
    FilterArgs a;
    int cpw_b, cpw_g, vpt, nblocks;
    size_t lds;
};
static int plan_filter(const LatticeView &Lg, const LatticeView &Lb, const MeanfieldBufs &buf, int B, int C, bool gauss_local,
                       FilterPlan &P) {
    if (C < 1 || C > kMaxLabels) return set_error(DSRG_ERR_UNSUPPORTED, "1 <= nlabels <= %d required", kMaxLabels);
    const int N = Lb.N;
    const int vs_b = (Lb.Mcap + 1 + 3) & ~3, vs_g = (Lg.Mcap + 1 + 3) & ~3;
    const size_t kLds = 157 * 1024;          // of the CU's 160 KB; the kernel has no static LDS
    auto lds_for = [&](int cpw_b, int cpw_g) {
        const size_t lb = (size_t)cpw_b * ((size_t)vs_b + N) * sizeof(float);
        const size_t lg = (size_t)cpw_g * ((size_t)vs_g + N) * sizeof(float);
        return lb > lg ? lb : lg;
    };
    // planes per block: bilateral 2 (8-byte LDS gathers) and Gaussian 4 (16-byte) when LDS holds them and
    // the batch is large enough to still fill the chip; otherwise narrower blocks
    int cpw_b = 1, cpw_g = 1;
    // (with the Gaussian workgroups out of the launch, one-plane bilateral workgroups still make ONE round of MI355X's 256 CUs
    // up to 12 images of 21 labels, and a one-plane workgroup is the shorter one: measured 14.1 / 14.3 us per launch at 8 / 12
    // images against 16.8-17.4 / 16.3-16.5 with plane pairs)
#if DSRG_EXP & 64
    const bool one_round_of_single_planes = gauss_local && (filter_opts() & kOptLocalGauss) && (size_t)B * C <= 512;
#else
    const bool one_round_of_single_planes = gauss_local && (filter_opts() & kOptLocalGauss) && (size_t)B * C <= 256;
#endif
    // (plane pairs only up to 16 vertices per thread: the 25-vertex instantiation with two planes has no registers left at 1 024
    // threads — 156 bytes of scratch per lane — where the one-plane one has none; maps of 52 .. 53 pixels a side)
    const bool pairs_fit_registers = (Lb.Mcap + kWG - 1) / kWG <= 16;
    if (lds_for(2, 4) <= kLds && (size_t)B * ((C + 1) / 2) >= 64 && C > 2 && !one_round_of_single_planes && pairs_fit_registers) { cpw_b = 2; cpw_g = 4; }
    else if (lds_for(1, 2) <= kLds && C > 2) { cpw_b = 1; cpw_g = 2; }
    else if (lds_for(1, 1) > 158 * 1024) return set_error(DSRG_ERR_UNSUPPORTED, "lattice does not fit LDS");
    const int vpt = (Lb.Mcap + kWG - 1) / kWG;
    const int ppt_tab = vpt <= 4 ? 1 : vpt <= 10 ? 2 : vpt <= 16 ? 3 : vpt <= 25 ? 5 : 6;
    if (N > ppt_tab * kWG) return set_error(DSRG_ERR_UNSUPPORTED, "pixel count %d exceeds the filter kernel", N);
    if (Lg.Mcap > ((vpt <= 4 ? 4 : vpt <= 10 ? 10 : vpt <= 16 ? 16 : vpt <= 25 ? 25 : 32) + 1) / 2 * kWG)
        return set_error(DSRG_ERR_UNSUPPORTED, "Gaussian lattice exceeds the filter kernel");
    FilterArgs &a = P.a;
    a.Lg = Lg; a.Lb = Lb; a.q = buf.q; a.msg_g = buf.msg_g; a.msg_b = buf.msg_b;
    a.B = B; a.C = C; a.N = N;
    a.groups_b = (C + cpw_b - 1) / cpw_b;
    // one workgroup per CU (LDS), 32 CUs per XCD, blockIdx % 8 picks the XCD: keep whole images on one XCD for the largest
    // multiple of 8 images (small batches are padded up to 8), spread the remaining images' blocks over all XCDs
    a.lat_stride = B < 8 ? 8 : (B & ~7);
    a.nblk_xcd = a.groups_b * a.lat_stride;
    a.nblk_b = a.nblk_xcd + (B > a.lat_stride ? a.groups_b * (B - a.lat_stride) : 0);
    // Gaussian blocks come last in the grid and fill CUs as bilateral blocks (twice as long) drain; none at all when the
    // host knows the lattice to be pixel-local (then the update kernel forms that message)
    a.groups_g = (C + cpw_g - 1) / cpw_g;
    a.opts = filter_opts();
    if (C <= 2) a.opts = (a.opts & ~kOptLocalGauss) | kOptSeq;     // the reference switches arithmetic at value_size <= 2
    const bool drop_gauss = gauss_local && (a.opts & kOptLocalGauss);
    a.nunits = a.nblk_b + (drop_gauss ? 0 : a.groups_g * B);
    a.dbg = reinterpret_cast<unsigned long long *>(g_filter_dbg);
    // enough LDS for two value buffers of the largest lattice when the CU has it (filter_lattice then blurs ping-pong for
    // every lattice whose actual vertex count fits), never less than one buffer + the input planes
    size_t lds = lds_for(cpw_b, cpw_g);
    {
        const size_t pb = (size_t)cpw_b * 2 * ((size_t)Lb.Mcap + 2) * sizeof(float), pg = (size_t)cpw_g * 2 * ((size_t)Lg.Mcap + 2) * sizeof(float);
        size_t want = pb > pg ? pb : pg;
        if (want > kLds) want = kLds;
        if (want > lds) lds = want;
        lds = (lds + 15) & ~(size_t)15;
    }
    a.lds_bytes = (int)lds;
    P.cpw_b = cpw_b; P.cpw_g = cpw_g; P.vpt = vpt; P.nblocks = a.nunits; P.lds = lds;
    return DSRG_OK;
}
// measurement (bench.py's LDS model must describe the launch that runs, not re-derive the launcher's choices): planes per
// bilateral / Gaussian workgroup, workgroups in the grid, dynamic LDS bytes of a filter launch over B images
int filter_plan_query(const LatticeView &Lg, const LatticeView &Lb, const MeanfieldBufs &buf, int B, int C, bool gauss_local,
                      int out[4]) {
    FilterPlan P;
    int rc = plan_filter(Lg, Lb, buf, B, C, gauss_local, P);
    if (rc) return rc;
    out[0] = (P.a.opts & kOptSeq) ? 1 : P.cpw_b;
    out[1] = (P.a.opts & kOptSeq) ? 1 : P.cpw_g;
    out[2] = P.nblocks;
    out[3] = (int)P.lds;
    return DSRG_OK;
}

static int run_filter(const FilterPlan &P, hipStream_t stream, Profiler *prof) {
    if (P.a.opts & kOptSeq) return dispatch_vpt<1, 1, true>(P.a, P.nblocks, P.lds, P.vpt, stream, prof);
    if (P.cpw_b == 2) return dispatch_vpt<2, 4>(P.a, P.nblocks, P.lds, P.vpt, stream, prof);
    if (P.cpw_g == 2) return dispatch_vpt<1, 2>(P.a, P.nblocks, P.lds, P.vpt, stream, prof);
    return dispatch_vpt<1, 1>(P.a, P.nblocks, P.lds, P.vpt, stream, prof);
}

int launch_meanfield(const LatticeView &Lg, const LatticeView &Lb, const MeanfieldBufs &buf, int B, int C,
                     const float *neg_unary, float wg, float wb, int n_iters, float *q_out,
                     double *refined_out, float *logq_out, bool gauss_local, hipStream_t stream, Profiler *prof, bool q0_ready) {
    FilterPlan P;
    int rc = plan_filter(Lg, Lb, buf, B, C, gauss_local, P);
    if (rc) return rc;
    UpdLocal loc;
    loc.la = reinterpret_cast<const float4 *>(Lg.loc_a); loc.lz = Lg.loc_z;
    loc.flags = (P.a.opts & kOptLocalGauss) ? Lg.flags : nullptr;
    const int N = Lb.N;
    // Q0 = expAndNormalize(-unary)   (densecrf.cpp:120); the fused step's softmax pass has written it already
    if (!(q0_ready && n_iters > 0)) {
        rc = launch_update(neg_unary, buf, loc, wg, wb, 0, n_iters > 0 ? buf.q : q_out,
                           n_iters > 0 ? nullptr : refined_out, n_iters > 0 ? nullptr : logq_out, B, C, N, stream);
        if (rc) return rc;
    }
    for (int it = 0; it < n_iters; it++) {
        rc = run_filter(P, stream, prof);
        if (rc) return rc;
        const bool last = (it == n_iters - 1);
        rc = launch_update(neg_unary, buf, loc, wg, wb, 1, last ? q_out : buf.q, last ? refined_out : nullptr,
                           last ? logq_out : nullptr, B, C, N, stream);
        if (rc) return rc;
    }
    return DSRG_OK;
}

// The normalisation vectors of nlat freshly built d = 5 lattices, norm = 1/sqrt(K 1 + 1e-20) (pairwise.cpp:44,54-57), as one
// launch of the filter kernel over a plane of ones: one workgroup per lattice, one plane, seqCompute arithmetic (value_size
// 1).  Replaces the build's own single-purpose pass for these lattices (23 -> 11 us at 41x41).
int launch_lattice_norm_pass(const LatticeView &L, int nlat, hipStream_t stream) {
    FilterArgs a;
    memset(&a, 0, sizeof(a));
    a.Lg = L; a.Lb = L;
    a.q = L.norm; a.msg_b = L.norm; a.msg_g = L.norm;     // q is never used (ones); the Gaussian half of the grid is empty
    a.B = nlat; a.C = 1; a.N = L.N;
    a.groups_b = 1; a.groups_g = 0;
    a.lat_stride = nlat < 8 ? 8 : (nlat & ~7);
    a.nblk_xcd = a.lat_stride;
    a.nblk_b = a.nblk_xcd + (nlat > a.lat_stride ? nlat - a.lat_stride : 0);
    a.nunits = a.nblk_b;
    a.opts = (filter_opts() & kOptSlotGuard) | kOptSeq | kOptNormPass;
    a.dbg = nullptr;
    const size_t kLds = 157 * 1024;
    const int vs = (L.Mcap + 1 + 3) & ~3;
    size_t lds = ((size_t)vs + L.N) * sizeof(float);
    size_t want = (size_t)2 * ((size_t)L.Mcap + 2) * sizeof(float);
    if (want > kLds) want = kLds;
    if (want > lds) lds = want;
    lds = (lds + 15) & ~(size_t)15;
    if (lds > 158 * 1024) return set_error(DSRG_ERR_UNSUPPORTED, "lattice does not fit LDS");
    a.lds_bytes = (int)lds;
    const int vpt = (L.Mcap + kWG - 1) / kWG;
    return dispatch_vpt<1, 1, true>(a, a.nunits, lds, vpt, stream, nullptr);
}

// One application of one normalised kernel (DenseKernel::filter, pairwise.cpp:63-80) to caller-supplied planes — the parity
// tests' view of splat / blur / slice without the softmax contraction of the inference loop.  kind 0 Gaussian, 1 bilateral.
int launch_filter_once(const LatticeView &Lg, const LatticeView &Lb, const MeanfieldBufs &buf, int B, int C, int kind,
                       const float *q_in, float *out, bool gauss_local, hipStream_t stream) {
    FilterPlan P;
    int rc = plan_filter(Lg, Lb, buf, B, C, false, P);
    if (rc) return rc;
    const int N = Lb.N;
    P.a.q = q_in;
    if (!gauss_local) P.a.opts &= ~kOptLocalGauss;      // the host has not seen the flag: the filter launch forms the message
    if (kind == 0) { P.a.msg_g = out; } else { P.a.msg_b = out; }
    rc = run_filter(P, stream, nullptr);
    if (rc) return rc;
    if (kind == 0 && gauss_local && (P.a.opts & kOptLocalGauss)) {
        // the filter launch skipped the Gaussian units (device-side flag test): form the message the way the update kernel does
        hipLaunchKernelGGL(gauss_local_kernel, dim3((N + 255) / 256, 32), dim3(256), 0, stream,
                           reinterpret_cast<const float4 *>(Lg.loc_a), Lg.loc_z, q_in, out, B * C, N);
        DSRG_LAUNCH_CHECK();
    }
    return DSRG_OK;
}

}  // namespace dsrg
