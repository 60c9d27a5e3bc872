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
// combines unary + weighted messages and renormalises over the labels.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "common.h"

namespace dsrg {

// ---------------------------------------------------------------------------------
// filter kernel: out_k[b][c][i] = norm_k[i] * (K_k (norm_k . Q[b][c]))[i]
// grid: one block per (label group, lattice); lattices of one image/kernel share an
// XCD (block id % 8) so their index arrays stay in one L2.
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
    unsigned int *counter;   // dynamic unit counter (zero at launch: the preceding update kernel resets it)
    int lds_val_stride;      // bilateral: vertices in the LDS value array (Mcap_b + 1, padded to 4)
    int lds_val_stride_g;    // Gaussian:  (Mcap_g + 1, padded to 4)
    int lds_bytes;           // dynamic LDS of the launch: value buffer(s) from the start, input planes [N] at the end
    unsigned long long *dbg; // optional per-workgroup phase timestamps (100 MHz wall clock), 2 x 16 per block
};

// What bounds this kernel (measured with per-phase timestamps, tools/filter_trace.py): not HBM and
// not LDS bandwidth but exposed memory latency — a dependent global load costs ~1.5 us here and a
// phase of LDS work only ~0.3 us.  Hence the structure:
//   * every index word a thread will need (splat entries, CSR row bounds, the neighbour words of
//     ALL blur axes, slice corners) is fetched up front / one phase ahead with unconditional SRSRC
//     buffer loads (per-lane offset in one VGPR, strides in the scalar offset, out-of-range reads
//     return 0) — one exposed round trip per lattice instead of one per phase;
//   * label planes are interleaved [vertex][CPW] in LDS and moved 8 bytes at a time (CPW = 2);
//   * one launch carries both lattices: bilateral blocks (2 label planes of one image, ~18 us) first in the grid,
//     Gaussian blocks (4 planes, ~9 us per image) after them so that they fill CUs as those drain; one workgroup per
//     CU (LDS), 176 + 96 blocks at B = 16 (see launch_meanfield for the block -> XCD map).
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

#define DSRG_STAMP(i_) do { if (dbg && tid == 0) dbg[(i_)] = wall_clock64(); } while (0)

// one lattice (dimension D, index li of set L), planes [c0, c0+nc) of image b:
//   out[c][i] = norm[i] * (K (norm . q[c]))[i]
// REG_IO (the persistent kernel): the marginals of this thread's pixels tid + p*kWG arrive in io[p][c] and the filtered
// values leave in io[p][c]; qb / out are not touched.
// `mid` runs once right after the lattice's index loads have been issued and before anything is consumed: the persistent
// kernel waits for the marginals there (and fills io), so the index round trip and the hand-off wait overlap.  M_pre /
// flags_pre: the lattice size and flags when the caller already holds them (REG_IO; saves a dependent load per call).
struct NoMid { __device__ __forceinline__ void operator()() const {} };
// the index words of one thread (splat entries, CSR rows, blur neighbours, slice corners, norm).  A caller that filters
// several plane groups through the SAME lattice (the Gaussian lattice is shared by all images) passes one of these and
// `reuse` = true from the second call on: nothing is fetched again (only possible when all D+1 axes fit the ring, D = 2).
template <int VPT, int PPT, int D> struct FilterIdx {
    static constexpr bool DEEP = VPT <= 10;
    static constexpr int KC = DEEP ? VPT : 8, RING = 3, D1 = D + 1;
    static constexpr bool kReusable = DEEP && D1 <= RING;
    float nrm[PPT];
    uint32_t epx[KC];
    float ew[KC];
    uint32_t rs0[KC], rs1[KC];
    uint32_t nbw[DEEP ? RING : 1][KC];
    uint32_t sv[PPT][D1];
    float sw[PPT][D1];
};
template <int CPW, int VPT, int PPT, int D, bool REG_IO = false, typename Mid = NoMid>
__device__ __forceinline__ void filter_lattice(const LatticeView &L, int li, const float *__restrict__ qb,
                                               float *__restrict__ out, int nc, int N,
                                               typename PlaneVec<CPW>::type *val,
                                               typename PlaneVec<CPW>::type *inq, unsigned long long *dbg,
                                               float (*io)[CPW], int tid_in, int M_pre, int flags_pre,
                                               Mid mid, int lds_elems, FilterIdx<VPT, PPT, D> &ix, bool reuse = false) {
    using vec_t = typename PlaneVec<CPW>::type;
    constexpr int D1 = D + 1;
    constexpr bool DEEP = VPT <= 10;             // all index words of a thread fit the register file
    constexpr int KC = DEEP ? VPT : 8;           // vertices per chunk of index loads otherwise
    constexpr int NCH = (VPT + KC - 1) / KC;
    if (!FilterIdx<VPT, PPT, D>::kReusable) reuse = false;
    float (&nrm)[PPT] = ix.nrm;
    uint32_t (&epx)[KC] = ix.epx;
    float (&ew)[KC] = ix.ew;
    uint32_t (&rs0)[KC] = ix.rs0;
    uint32_t (&rs1)[KC] = ix.rs1;
    uint32_t (&nbw)[DEEP ? 3 : 1][KC] = ix.nbw;
    uint32_t (&sv)[PPT][D1] = ix.sv;
    float (&sw)[PPT][D1] = ix.sw;
    // inside the persistent kernel's iteration loop the thread index arrives laundered (tid_in): everything derived from
    // it is then recomputed per iteration instead of being hoisted out of the loop and held in registers across it
    const int tid = REG_IO ? tid_in : (int)threadIdx.x;
    const int Mcap = L.Mcap, E = N * D1;
    const float *norm = L.norm + (size_t)li * N;

    const rsrc_t r_rs = make_rsrc(L.row_start + (size_t)li * (Mcap + 1), sizeof(uint32_t) * (size_t)(Mcap + 1));
    const rsrc_t r_nb = make_rsrc(L.nb + (size_t)li * D1 * Mcap, sizeof(uint32_t) * (size_t)D1 * Mcap);
    const rsrc_t r_cp = make_rsrc(L.csr_pix + (size_t)li * D1 * N, sizeof(uint16_t) * (size_t)D1 * N);
    const rsrc_t r_cw = make_rsrc(L.csr_w + (size_t)li * D1 * N, sizeof(float) * (size_t)D1 * N);
    const rsrc_t r_vid = make_rsrc(L.vid + (size_t)li * D1 * N, sizeof(uint16_t) * (size_t)D1 * N);
    const rsrc_t r_bary = make_rsrc(L.bary + (size_t)li * D1 * N, sizeof(float) * (size_t)D1 * N);
    const rsrc_t r_norm = make_rsrc(norm, sizeof(float) * (size_t)N);
    const rsrc_t r_q = make_rsrc(qb, sizeof(float) * (size_t)nc * N);

    const int M = REG_IO ? M_pre : L.M[li];
    const int lat_flags = REG_IO ? flags_pre : L.flags[li];
    DSRG_STAMP(0);

    if (lat_flags & 1) {
        // Diagonal lattice (e.g. the Gaussian kernel at training scale: sigma = 0.25 px):
        // every simplex corner is private to its pixel and has no blur neighbour, so splat, blur and
        // slice collapse to per-pixel arithmetic — evaluated here in the general path's operation
        // order (products, 0 + p, val + 0.5*(0+0), ordered slice sum), hence bit-identical to it.
        const float alpha = 1.0f / (1.0f + exp2f(-(float)D));
        float nvs[PPT], bws[PPT][D1];
#pragma unroll
        for (int p = 0; p < PPT; p++) {
            nvs[p] = ld_f32(r_norm, (uint32_t)tid * 4u, (uint32_t)p * (kWG * 4u));
#pragma unroll
            for (int r = 0; r < D1; r++)
                bws[p][r] = ld_f32(r_bary, (uint32_t)tid * 4u, ((uint32_t)p * kWG + (uint32_t)r * (uint32_t)N) * 4u);
        }
        mid();
#pragma unroll
        for (int p = 0; p < PPT; p++) {
            const int i = tid + p * kWG;
            const float nv = nvs[p];
            float bw[D1], qq[CPW];
#pragma unroll
            for (int r = 0; r < D1; r++) bw[r] = bws[p][r];
#pragma unroll
            for (int c = 0; c < CPW; c++) {
                if constexpr (REG_IO) qq[c] = io[p][c];
                else qq[c] = ld_f32(r_q, (uint32_t)tid * 4u, (uint32_t)p * (kWG * 4u) + (uint32_t)c * (uint32_t)N * 4u);
            }
            if (i < N) {
#pragma unroll
                for (int c = 0; c < CPW; c++) {
                    if (c < nc) {
                        const float x = qq[c] * nv;
                        float acc = 0.0f;
#pragma unroll
                        for (int r = 0; r < D1; r++) {
                            float v = 0.0f + bw[r] * x;          // splat into an empty vertex
                            v = v + 0.5f * (0.0f + 0.0f);        // d+1 blur passes without neighbours
                            acc = acc + (bw[r] * alpha) * v;     // slice
                        }
                        if constexpr (REG_IO) io[p][c] = acc * nv;
                        else out[(size_t)c * N + i] = acc * nv;
                    }
                }
            }
        }
        DSRG_STAMP(11);
        if (dbg && tid == 0) dbg[12] = (unsigned long long)M;
        return;
    }

    // ---- stage A: everything that does not depend on anything, in one burst
    float qv[PPT][CPW];
#pragma unroll
    for (int p = 0; p < PPT; p++) {
        if (!reuse) nrm[p] = ld_f32(r_norm, (uint32_t)tid * 4u, (uint32_t)p * (kWG * 4u));
        if constexpr (!REG_IO) {
#pragma unroll
            for (int c = 0; c < CPW; c++)
                qv[p][c] = ld_f32(r_q, (uint32_t)tid * 4u, (uint32_t)p * (kWG * 4u) + (uint32_t)c * (uint32_t)N * 4u);
        }
    }
    if (!reuse) {
#pragma unroll
        for (int k = 0; k < KC; k++) {             // splat entries e_k = tid + k*1024 and CSR rows of v_k
            epx[k] = ld_u16(r_cp, (uint32_t)tid * 2u, (uint32_t)k * (kWG * 2u));
            ew[k] = ld_f32(r_cw, (uint32_t)tid * 4u, (uint32_t)k * (kWG * 4u));
            rs0[k] = ld_u32(r_rs, (uint32_t)tid * 4u, (uint32_t)k * (kWG * 4u));
        }
        // row end = the next vertex's row start: the neighbouring lane holds it; only the last lane of a wave loads it
        if ((tid & 63) == 63) {
#pragma unroll
            for (int k = 0; k < KC; k++) rs1[k] = ld_u32(r_rs, (uint32_t)tid * 4u, (uint32_t)k * (kWG * 4u) + 4u);
        }
    }
    // neighbour words n1 | n2<<16 of my vertices: a ring of RING axes, fetched RING-1 passes ahead of
    // their use (a blur pass is shorter than one memory round trip)
    constexpr int RING = 3;
    auto load_axis = [&](int j) {
        if (reuse) return;
#pragma unroll
        for (int k = 0; k < KC; k++)
            nbw[j % RING][k] = ld_u32(r_nb, (uint32_t)tid * 4u, (uint32_t)j * (uint32_t)Mcap * 4u + (uint32_t)k * (kWG * 4u));
    };
    if constexpr (DEEP) load_axis(0);
    if (!reuse) {
#pragma unroll
        for (int k = 0; k < KC; k++) {
            const uint32_t up = __shfl_down(rs0[k], 1, 64);
            if ((tid & 63) != 63) rs1[k] = up;
        }
    }
    mid();
    if constexpr (REG_IO) {
#pragma unroll
        for (int p = 0; p < PPT; p++)
#pragma unroll
            for (int c = 0; c < CPW; c++) qv[p][c] = io[p][c];
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
    __syncthreads();
    DSRG_STAMP(1);

    // ---- splat (permutohedral.cpp:545-553), two steps that keep the reference's accumulation order
    // without a dependent global load: (1) entry-parallel products w_e * in[pixel_e] into LDS (the
    // entries are sorted by vertex, then by the reference's visiting order); (2) vertex-parallel
    // ordered sums over each vertex's contiguous row of products.  `prod` aliases `val`: the sums
    // wait in registers until every row has been read.
    vec_t *prod = val;
    vec_t sacc[VPT];
    const bool single = DEEP && (lat_flags & 4);
    if (single) {
        // every vertex has exactly one contributor (M == E; e.g. the Gaussian lattice at training scale): entry e is vertex
        // e's only term, so the splat is values[v] = 0 + w_v * in[pixel_v] — no products pass, no row sums
#pragma unroll
        for (int k = 0; k < KC; k++) {
            const vec_t x = inq[min((int)epx[k], N - 1)];
#pragma unroll
            for (int c = 0; c < CPW; c++) pv_set(sacc[k], c, 0.0f + ew[k] * pv_get(x, c));
        }
        if constexpr (DEEP) { if (1 < D1) load_axis(1); }
        if constexpr (DEEP) { if (2 < D1) load_axis(2); }
        DSRG_STAMP(2);
        DSRG_STAMP(3);
    } else {
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
        if (ch > 0) {
#pragma unroll
            for (int k = 0; k < KC; k++) {
                epx[k] = ld_u16(r_cp, (uint32_t)tid * 2u, (uint32_t)(ch * KC + k) * (kWG * 2u));
                ew[k] = ld_f32(r_cw, (uint32_t)tid * 4u, (uint32_t)(ch * KC + k) * (kWG * 4u));
            }
        }
#pragma unroll
        for (int k = 0; k < KC; k++) {
            const int e = tid + (ch * KC + k) * kWG;
            if (e < E) {
                const vec_t x = inq[min((int)epx[k], N - 1)];
                vec_t p;
#pragma unroll
                for (int c = 0; c < CPW; c++) pv_set(p, c, ew[k] * pv_get(x, c));
                prod[e] = p;
            }
        }
    }
    if constexpr (DEEP) { if (1 < D1) load_axis(1); }   // the entry registers are free now
    __syncthreads();
    DSRG_STAMP(2);
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
        if (ch > 0) {
#pragma unroll
            for (int k = 0; k < KC; k++) {
                const uint32_t so = (uint32_t)(ch * KC + k) * (kWG * 4u);
                rs0[k] = ld_u32(r_rs, (uint32_t)tid * 4u, so);
                rs1[k] = ld_u32(r_rs, (uint32_t)tid * 4u, so + 4u);
            }
        }
#pragma unroll
        for (int k = 0; k < KC; k++) {
            if (ch * KC + k < VPT) {
                const int v = tid + (ch * KC + k) * kWG;
                float s[CPW];
#pragma unroll
                for (int c = 0; c < CPW; c++) s[c] = 0.0f;
                const uint32_t t1 = (v < M) ? rs1[k] : rs0[k];
                for (uint32_t t = rs0[k]; t < t1; t++) {
                    const vec_t p = prod[t];
#pragma unroll
                    for (int c = 0; c < CPW; c++) s[c] = s[c] + pv_get(p, c);
                }
#pragma unroll
                for (int c = 0; c < CPW; c++) pv_set(sacc[ch * KC + k], c, s[c]);
            }
        }
    }
    DSRG_STAMP(3);
    if constexpr (DEEP) { if (2 < D1) load_axis(2); }
    }
    __syncthreads();                                     // every row of products (or every input value) has been consumed
    // two value buffers when the region holds them (the input planes are dead by now): an axis then gathers from one and
    // writes the other — one barrier per axis instead of two
    const int vstride = (M + 2) & ~1;
    const bool pingpong = 2 * vstride <= lds_elems;
    vec_t *cur = val, *nxt = pingpong ? val + vstride : val;
#pragma unroll
    for (int k = 0; k < VPT; k++) {
        const int v = tid + k * kWG;
        if (v < M) cur[v] = sacc[k];
    }
    if (tid == 0) {                                     // zero sentinel = "no neighbour" (permutohedral.cpp:561-562): slot M
        vec_t z;
#pragma unroll
        for (int c = 0; c < CPW; c++) pv_set(z, c, 0.0f);
        cur[M] = z;
        nxt[M] = z;
    }
    __syncthreads();
    DSRG_STAMP(4);

    // ---- blur along the d+1 lattice axes (permutohedral.cpp:556-569): Jacobi per axis — new values
    // held in registers between the read barrier and the write barrier
#pragma unroll
    for (int j = 0; j < D1; j++) {
        // sacc[k] holds the current value of my vertex v_k (no LDS read for it)
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            if constexpr (!DEEP) {
#pragma unroll
                for (int k = 0; k < KC; k++)
                    nbw[0][k] = ld_u32(r_nb, (uint32_t)tid * 4u,
                                       (uint32_t)j * (uint32_t)Mcap * 4u + (uint32_t)(ch * KC + k) * (kWG * 4u));
            }
#pragma unroll
            for (int k = 0; k < KC; k++) {
                if (ch * KC + k < VPT) {
                    // (the words of the unused tail v >= M point at the zero sentinel: no test needed)
                    const uint32_t word = nbw[DEEP ? j % RING : 0][k];
                    const int n1 = (int)(word & 0xffffu), n2 = (int)(word >> 16);
                    const vec_t x1 = cur[n1], x2 = cur[n2];
#pragma unroll
                    for (int c = 0; c < CPW; c++) {
                        float s = pv_get(x1, c) + pv_get(x2, c);
                        s = 0.5f * s;
                        pv_set(sacc[ch * KC + k], c, pv_get(sacc[ch * KC + k], c) + s);
                    }
                    if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // bound the LDS gathers in flight
                }
            }
        }
        if constexpr (DEEP) { if (j + RING < D1) load_axis(j + RING); }     // this axis' ring slot is free
        if (j == (D1 >= 3 ? D1 - 3 : 0) && !reuse) {   // slice corners, two passes ahead of their use
#pragma unroll
            for (int p = 0; p < PPT; p++) {
#pragma unroll
                for (int r = 0; r < D1; r++) {
                    const uint32_t so = (uint32_t)p * kWG + (uint32_t)r * (uint32_t)N;
                    sv[p][r] = ld_u16(r_vid, (uint32_t)tid * 2u, so * 2u);
                    sw[p][r] = ld_f32(r_bary, (uint32_t)tid * 4u, so * 4u);
                }
            }
        }
        if (pingpong) {
#pragma unroll
            for (int k = 0; k < VPT; k++) {
                const int v = tid + k * kWG;
                if (v < M) nxt[v] = sacc[k];
            }
            __syncthreads();                         // gathers of this axis done, values of the next one in place
            vec_t *t = cur; cur = nxt; nxt = t;
        } else {
            __syncthreads();                         // all gathers of this axis are done
#pragma unroll
            for (int k = 0; k < VPT; k++) {
                const int v = tid + k * kWG;
                if (v < M) cur[v] = sacc[k];
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
                const float w = sw[p][r] * alpha;
                const vec_t x = cur[min(sv[p][r], (uint32_t)M)];
#pragma unroll
                for (int c = 0; c < CPW; c++) acc[c] = acc[c] + w * pv_get(x, c);
            }
#pragma unroll
            for (int c = 0; c < CPW; c++) {
                if constexpr (REG_IO) io[p][c] = acc[c] * nrm[p];
                else if (c < nc) out[(size_t)c * N + i] = acc[c] * nrm[p];
            }
        }
    }
    DSRG_STAMP(11);
    if (dbg && tid == 0) dbg[12] = (unsigned long long)M;
}

// One launch filters every label plane of every image through both lattices.  Work units: first the bilateral ones
// (CPW_B planes of one image: 6 axes, M ~ 2-6 N vertices, the long ones), then the Gaussian ones (CPW_G planes of one image
// through the lattice all images share: 3 axes).  A workgroup starts with unit blockIdx.x and then pulls further units from
// a counter (zeroed by the update kernel that precedes every filter launch), so that the CUs the bilateral units leave
// idle (80 of 256 at 16 images) work through the Gaussian units and nothing queues behind a long unit; a workgroup that
// runs several Gaussian units keeps the lattice's index words in registers between them.
// Keeping the Gaussian lattice's index words in registers across a workgroup's Gaussian units was measured and lost: the
// extra live registers slowed the first unit from 9.0 to 11.3 us (B = 16), more than the second unit gained.
constexpr bool kKeepGaussIdx = false;
template <int CPW_B, int CPW_G, int VPT_B, int PPT>
__global__ __launch_bounds__(kWG) void mf_filter_kernel(FilterArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int VPT_G = (VPT_B + 1) / 2;                 // Mcap_gauss = Mcap_bilateral / 2
    unsigned long long *dbg = a.dbg ? a.dbg + (size_t)blockIdx.x * 32 : nullptr;
    int *slot = reinterpret_cast<int *>(smem + a.lds_bytes);           // next-unit broadcast (16 bytes past the region)
    // the next unit of this workgroup (workgroup-uniform); units are handed out in increasing order, so a workgroup sees its
    // bilateral units first and its Gaussian units after them
    auto next_unit = [&]() -> int {
        if (a.nunits <= (int)gridDim.x) return a.nunits;                   // every unit was assigned statically
        __syncthreads();                                                   // LDS is reused by the next unit
        if (threadIdx.x == 0) {
            // look before taking a ticket: the workgroups that finish the long units together would otherwise queue on the
            // counter only to learn that nothing is left
            int next = a.nunits;
            if ((int)gridDim.x + (int)__hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.nunits)
                next = (int)gridDim.x + (int)atomicAdd(a.counter, 1u);
            *slot = next;
        }
        __syncthreads();
        return *slot;
    };
    int unit = blockIdx.x;
    while (unit < a.nblk_b) {
        using vec_t = typename PlaneVec<CPW_B>::type;
        // units of one image share unit % 8, i.e. (for the statically assigned ones) one XCD and its L2; the images beyond
        // the last multiple of 8 (B = 20: images 16..19) are laid out image-major so that they spread evenly over the XCDs
        int g, b;
        if (unit < a.nblk_xcd) { g = unit / a.lat_stride; b = unit % a.lat_stride; }
        else { const int rel = unit - a.nblk_xcd; b = a.lat_stride + rel / a.groups_b; g = rel % a.groups_b; }
        if (b < a.B) {
            const int c0 = g * CPW_B, nc = min(CPW_B, a.C - c0);
            vec_t *val = reinterpret_cast<vec_t *>(smem);                      // value buffer(s), label-interleaved
            vec_t *inq = reinterpret_cast<vec_t *>(smem + a.lds_bytes) - a.N;  // [N] at the end of the region
            const size_t o = ((size_t)b * a.C + c0) * a.N;
            FilterIdx<VPT_B, PPT, 5> bix;
            filter_lattice<CPW_B, VPT_B, PPT, 5>(a.Lb, b, a.q + o, a.msg_b + o, nc, a.N, val, inq,
                                                 unit == (int)blockIdx.x ? dbg : nullptr, nullptr, 0, 0, 0, NoMid(),
                                                 a.lds_bytes / (int)sizeof(vec_t), bix);
        }
        unit = next_unit();
    }
    {
        using vec_t = typename PlaneVec<CPW_G>::type;
        FilterIdx<VPT_G, PPT, 2> gix;                                          // kept across this workgroup's Gaussian units
        bool g_loaded = false;
        while (unit < a.nunits) {
            const int rel = unit - a.nblk_b;
            const int b = rel / a.groups_g, g = rel % a.groups_g;
            const int c0 = g * CPW_G, nc = min(CPW_G, a.C - c0);
            vec_t *val = reinterpret_cast<vec_t *>(smem);
            vec_t *inq = reinterpret_cast<vec_t *>(smem + a.lds_bytes) - a.N;
            const size_t o = ((size_t)b * a.C + c0) * a.N;
            filter_lattice<CPW_G, VPT_G, PPT, 2>(a.Lg, 0, a.q + o, a.msg_g + o, nc, a.N, val, inq,
                                                 (dbg && unit == (int)blockIdx.x) ? dbg + 16 : nullptr, nullptr, 0, 0, 0, NoMid(),
                                                 a.lds_bytes / (int)sizeof(vec_t), gix, g_loaded);
            g_loaded = kKeepGaussIdx;
            unit = next_unit();
        }
    }
}
#undef DSRG_STAMP

// ---------------------------------------------------------------------------------
// per-pixel update: Q = expAndNormalize( -U - sum_k (-w_k msg_k) )   (densecrf.cpp:98-106,122-128)
// The last iteration additionally emits the layer outputs of pylayers.py:84-88.
// numpy's float64 add-reduce over a contiguous run of n <= CT values (8 partial sums, then the
// tail) — the order np.sum(result, axis=1) uses for the label axis (pylayers.py:86,330).
// Written with static indices only so the column stays in registers for a run-time n.
template <int CT> __device__ __forceinline__ double np_pairwise_sum(const double (&a)[CT], int n) {
    if (n < 8) {
        double r = 0.0;
#pragma unroll
        for (int i = 0; i < (CT < 8 ? CT : 8); i++)
            if (i < n) r += a[i];
        return r;
    }
    if constexpr (CT >= 8) {
        const int full = n - (n % 8);
        double r[8];
#pragma unroll
        for (int k = 0; k < 8; k++) r[k] = a[k];
#pragma unroll
        for (int i = 8; i < CT; i++)
            if (i < full) r[i & 7] += a[i];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
        for (int i = 8; i < CT; i++)
            if (i >= full && i < n) res += a[i];
        return res;
    }
    return 0.0;
}

template <int CT, bool USE_MSGS>   // CT = compile-time bound on C (loops fully unrolled, values in registers)
__global__ __launch_bounds__(256) void mf_update_kernel(const float *__restrict__ neg_unary,
                                                        const float *__restrict__ msg_g,
                                                        const float *__restrict__ msg_b, float wg, float wb,
                                                        float *__restrict__ q_out,
                                                        double *__restrict__ refined_out,
                                                        float *__restrict__ logq_out, int B, int C, int N,
                                                        unsigned int *__restrict__ work_counter) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0 && work_counter) *work_counter = 0u;          // the filter launch that follows starts its unit queue at 0
    if (idx >= B * N) return;
    const int b = idx / N, i = idx - b * N;
    const size_t base = (size_t)b * C * N + i;
    // all loads first, unconditionally (label index clamped), so they are in flight together
    float t[CT], mg[CT], mb[CT];
#pragma unroll
    for (int c = 0; c < CT; c++) {
        const size_t o = base + (size_t)min(c, C - 1) * N;
        t[c] = neg_unary[o];                                         // tmp1 = -unary
        if (USE_MSGS) { mg[c] = msg_g[o]; mb[c] = msg_b[o]; }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < CT; c++) {
        float v = t[c];
        if (USE_MSGS) {
            // tmp2 = -w * filter(Q); tmp1 -= tmp2  — Gaussian first, then bilateral
            const float m1 = (-wg) * mg[c];
            v = v - m1;
            const float m2 = (-wb) * mb[c];
            v = v - m2;
        }
        t[c] = (c < C) ? v : -INFINITY;
        mx = fmaxf(mx, t[c]);
    }
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < CT; c++) {
        const float e = exp_cr(t[c] - mx);
        t[c] = e;
        sum = (c < C) ? sum + e : sum;
    }
#pragma unroll
    for (int c = 0; c < CT; c++) t[c] = t[c] / sum;
    if (q_out) {
#pragma unroll
        for (int c = 0; c < CT; c++)
            if (c < C) q_out[base + (size_t)c * N] = t[c];
    }
    if (refined_out) {
        // pylayers.py:84-88: float64, clip at min_prob, divide by the label sum, log
        double col[CT];
#pragma unroll
        for (int c = 0; c < CT; c++) {
            double v = (c < C) ? (double)t[c] : 0.0;
            col[c] = v < 0.0001 ? 0.0001 : v;
        }
        const double s = np_pairwise_sum<CT>(col, C);
#pragma unroll
        for (int c = 0; c < CT; c++) {
            if (c < C) {
                const double r = col[c] / s;
                refined_out[base + (size_t)c * N] = r;
                if (logq_out) logq_out[base + (size_t)c * N] = (float)log(r);
            }
        }
    }
}

// The same update with kUpdParts threads per pixel: lanes = 64 consecutive pixels (coalesced plane loads), wave p of the
// workgroup takes the labels c = p, p + kUpdParts, ...  The ~1.4 us chain of 21 dependent fp64 exps per thread of the kernel
// above (it runs at less than one wave per SIMD: nothing hides it) becomes ceil(C / kUpdParts) exps; the column max, the
// label-order sum and numpy's pairwise sum go through LDS in exactly the reference's order, so the results are bit-identical.
constexpr int kUpdParts = 4, kUpdPix = 64;
template <int CT, bool USE_MSGS>   // CT = compile-time bound on C
__global__ __launch_bounds__(kUpdParts * kUpdPix) void mf_update_split_kernel(
    const float *__restrict__ neg_unary, const float *__restrict__ msg_g, const float *__restrict__ msg_b, float wg, float wb,
    float *__restrict__ q_out, double *__restrict__ refined_out, float *__restrict__ logq_out, int B, int C, int N,
    unsigned int *__restrict__ work_counter) {
    __shared__ float ev[CT][kUpdPix];                      // e, then q per (label, pixel)
    __shared__ float pm[kUpdParts][kUpdPix];               // partial column maxima
    const int px = threadIdx.x & (kUpdPix - 1), part = threadIdx.x >> 6;
    const int idx = blockIdx.x * kUpdPix + px;
    if (blockIdx.x == 0 && threadIdx.x == 0 && work_counter) *work_counter = 0u;
    const bool live = idx < B * N;
    const int b = live ? idx / N : 0, i = live ? idx - b * N : 0;
    const size_t base = (size_t)b * C * N + i;
    constexpr int LPT = (CT + kUpdParts - 1) / kUpdParts;
    float t[LPT], mg[LPT], mb[LPT];
#pragma unroll
    for (int k = 0; k < LPT; k++) {                        // all loads first, unconditionally (label index clamped)
        const size_t o = base + (size_t)min(part + k * kUpdParts, C - 1) * N;
        t[k] = neg_unary[o];
        if (USE_MSGS) { mg[k] = msg_g[o]; mb[k] = msg_b[o]; }
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
    mx = fmaxf(fmaxf(pm[0][px], pm[1][px]), fmaxf(pm[2][px], pm[3][px]));      // fmax is order-independent
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
// The whole inference loop (densecrf.cpp:115-131) as ONE launch: workgroup (b, g) keeps plane group g of image b for all
// iterations.  An iteration has a plane-parallel half (the two filters on my planes, for every pixel) and a pixel-parallel
// half (expAndNormalize needs all labels of a pixel: workgroup g owns the pixel range g of its image), so the marginals
// are transposed twice per iteration between the G workgroups of an image.  The hand-off is the data itself: 8-byte
// {tag, value} granules, one relaxed agent-scope (write-through, L1-bypassing) store each, polled by their reader until
// the tag of the expected phase shows (cdna_hip_programming.md, Guideline 16, form R2) — no flag, no fence, no grid
// barrier, and images never wait for one another.  tag = launch epoch << 6 | iteration: a granule is written exactly once
// per (launch, iteration), and the protocol cannot overwrite one before its readers are through (a writer needs every
// reader's next output first).  All G workgroups of an image must be resident: the host launches at most 256 of them.
struct PersistArgs {
    LatticeView Lg, Lb;
    const float *neg_unary;            // (B,C,N)  -U
    unsigned long long *qg, *vg;       // (B,C,N) granules: marginals Q, pre-normalisation values V = -U - sum_k (-w_k msg_k)
    float *q_out;                      // (B,C,N) marginals of the last iteration (may be null)
    double *refined_out;               // (B,C,N) pylayers.py:84-86 (may be null)
    float *logq_out;                   // (B,C,N) pylayers.py:88 (may be null)
    unsigned int *status;              // host-mapped word: set when a hand-off timed out
    float wg, wb;
    int b0, B, C, N, n_iters;          // this launch covers images [b0, B)
    int nblk_xcd, lat_stride, groups;  // block -> (group, image - b0) map, as in FilterArgs
    int lds_val_stride;                // vertices in the LDS value array (bilateral Mcap + 1, padded to 4)
    int lp_shift;                      // log2 of the lanes per pixel in the pixel-parallel half (5: C <= 32, 6: C <= 64)
    unsigned int epoch;
    unsigned long long *dbg;           // tools: 64 phase timestamps (100 MHz wall clock) per workgroup
};

constexpr unsigned int kSpinLimit = 1u << 18;

__device__ __forceinline__ void store_granule(unsigned long long *g, unsigned int tag, float v) {
    __hip_atomic_store(g, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
// every lane polls its K granules (null = nothing to wait for) until all lanes of the wave hold the expected tag
template <int K>
__device__ __forceinline__ void poll_granules(const unsigned long long *const (&g)[K], unsigned int tag, float (&v)[K],
                                              unsigned int *status) {
    unsigned int need = 0;
#pragma unroll
    for (int k = 0; k < K; k++) { v[k] = 0.0f; if (g[k]) need |= 1u << k; }
    for (unsigned int spins = 0;; spins++) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            if ((need >> k) & 1u) {
                const unsigned long long x = __hip_atomic_load(g[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned int)(x >> 32) == tag) { v[k] = __uint_as_float((unsigned int)x); need &= ~(1u << k); }
            }
        }
        if (__all(need == 0)) return;
        if (spins >= kSpinLimit || (spins > 64 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM))) {
            __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // bounded: never hang the GPU
            return;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// pixel-parallel half for my pixel range [px0, px1): LP lanes per pixel, lane = label, kWG / LP pixels per round.  USE_V:
// values from the V granules of iteration `it`, else from -U (Q0, densecrf.cpp:120).  Not last: Q granules of iteration
// `it`; last: the layer outputs.  The granules of kUpdBatch rounds are requested together (one memory round trip).
constexpr int kUpdBatch = 8;
template <bool USE_V>
__device__ __forceinline__ void update_phase(const PersistArgs &a, int b, int px0, int px1, int it, bool last, float *scr) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));          // nothing derived from it is hoisted out of the iteration loop (register pressure)
    const int C = a.C, N = a.N, lp = 1 << a.lp_shift, ppr = kWG >> a.lp_shift;
    const int c = tid & (lp - 1), grp = tid & ~(lp - 1);
    const size_t img = (size_t)b * C * N + (size_t)min(c, C - 1) * N;
    const unsigned int tag = (a.epoch << 6) | (unsigned int)it;
    for (int base0 = px0; base0 < px1; base0 += kUpdBatch * ppr) {
        float tv[kUpdBatch];
        {
            const unsigned long long *gp[kUpdBatch];
#pragma unroll
            for (int r = 0; r < kUpdBatch; r++) {
                const int i = base0 + r * ppr + (tid >> a.lp_shift);
                const bool act = i < px1 && c < C;
                if (USE_V) gp[r] = act ? a.vg + img + i : nullptr;
                else tv[r] = a.neg_unary[img + min(i, N - 1)];
            }
            if (USE_V) poll_granules<kUpdBatch>(gp, tag, tv, a.status);
        }
#pragma unroll
        for (int r = 0; r < kUpdBatch; r++) {
            if (base0 + r * ppr >= px1) break;                   // workgroup-uniform
            const int i = base0 + r * ppr + (tid >> a.lp_shift);
            const bool act = i < px1 && c < C;
            const size_t o = img + min(i, N - 1);
            const float t = act ? tv[r] : -INFINITY;
            float mx = t;                                        // column max (densecrf.cpp:101); fmax is order-independent
            for (int off = lp >> 1; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
            const float e = act ? exp_cr(t - mx) : 0.0f;
            __builtin_amdgcn_wave_barrier();
            scr[tid] = e;                                        // read back only by lanes of this wave: no workgroup barrier
            __builtin_amdgcn_wave_barrier();
            float sum = 0.0f;                                    // label-order sum, as the reference's column sum
            for (int cc = 0; cc < C; cc++) sum = sum + scr[grp + cc];
            const float q = e / sum;
            if (!last) {
                if (act) store_granule(a.qg + o, tag, q);
            } else {
                if (act && a.q_out) a.q_out[o] = q;
                if (a.refined_out) {
                    // pylayers.py:84-88: float64, clip at min_prob, divide by numpy's pairwise label sum, log
                    __builtin_amdgcn_wave_barrier();
                    scr[tid] = q;
                    __builtin_amdgcn_wave_barrier();
                    auto col = [&](int k) { const double v = (double)scr[grp + k]; return v < 0.0001 ? 0.0001 : v; };
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
                    if (act) {
                        const double qd = (double)q;
                        const double rr = (qd < 0.0001 ? 0.0001 : qd) / s;
                        a.refined_out[o] = rr;
                        if (a.logq_out) a.logq_out[o] = (float)log(rr);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int CPW, int VPT_B, int PPT>
__global__ __launch_bounds__(kWG) void mf_persistent_kernel(PersistArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using vec_t = typename PlaneVec<CPW>::type;
    constexpr int VPT_G = (VPT_B + 1) / 2;                 // Mcap_gauss = Mcap_bilateral / 2
    int g, b;
    if ((int)blockIdx.x < a.nblk_xcd) { g = blockIdx.x / a.lat_stride; b = blockIdx.x % a.lat_stride; }
    else { const int rel = blockIdx.x - a.nblk_xcd; b = a.lat_stride + rel / a.groups; g = rel % a.groups; }
    b += a.b0;
    if (b >= a.B) return;
    const int tid = threadIdx.x, C = a.C, N = a.N;
    const int c0 = g * CPW, nc = min(CPW, C - c0);
    vec_t *val = reinterpret_cast<vec_t *>(smem);                      // [Mcap + 1] label-interleaved
    vec_t *inq = val + a.lds_val_stride;                               // [N]
    float *scr = reinterpret_cast<float *>(inq + N);                   // [kWG] scratch of the pixel-parallel half
    const int npx = (N + a.groups - 1) / a.groups, px0 = min(N, g * npx), px1 = min(N, px0 + npx);
    const size_t img = (size_t)b * C * N;

    float *stash = scr + kWG;                                          // [PPT * CPW][kWG] per-thread values parked in LDS
    unsigned long long *dbg = a.dbg ? a.dbg + (size_t)blockIdx.x * 64 : nullptr;
#define DSRG_PSTAMP(i_) do { if (dbg && threadIdx.x == 0 && (i_) < 64) dbg[(i_)] = wall_clock64(); } while (0)
    DSRG_PSTAMP(0);
    const rsrc_t r_u = make_rsrc(a.neg_unary + img + (size_t)c0 * N, sizeof(float) * (size_t)nc * N);
    const int Mg = a.Lg.M[0], flags_g = a.Lg.flags[0], Mb = a.Lb.M[b], flags_b = a.Lb.flags[b];   // loop invariants
    update_phase<false>(a, b, px0, px1, 0, a.n_iters == 0, scr);       // Q0 = expAndNormalize(-U)
    DSRG_PSTAMP(1);
    for (int it = 1; it <= a.n_iters; it++) {
        const unsigned int tag_q = (a.epoch << 6) | (unsigned int)(it - 1), tag_v = (a.epoch << 6) | (unsigned int)it;
        int tl = tid;
        asm volatile("" : "+v"(tl));                                   // see filter_lattice: no hoisting across iterations
        float io[PPT][CPW];
        // wait for the marginals of my planes, run the Gaussian kernel on them and fold it into
        // tmp1 = -U - (-w_g msg_g) (densecrf.cpp:122-127: Gaussian first, then bilateral); tmp1 is parked in LDS during the
        // bilateral filter and io holds the marginals again on return
        auto fetch_and_gauss = [&](auto gauss) {
            float pu[PPT][CPW];                                        // -U of my planes at my pixels
#pragma unroll
            for (int p = 0; p < PPT; p++)
#pragma unroll
                for (int c = 0; c < CPW; c++)
                    pu[p][c] = ld_f32(r_u, (uint32_t)tl * 4u, (uint32_t)p * (kWG * 4u) + (uint32_t)c * (uint32_t)N * 4u);
            const unsigned long long *gp[PPT * CPW];
            float qv[PPT * CPW];
#pragma unroll
            for (int p = 0; p < PPT; p++)
#pragma unroll
                for (int c = 0; c < CPW; c++) {
                    const int i = tl + p * kWG;
                    gp[p * CPW + c] = (i < N && c < nc) ? a.qg + img + (size_t)(c0 + c) * N + i : nullptr;
                }
            poll_granules<PPT * CPW>(gp, tag_q, qv, a.status);
            DSRG_PSTAMP(2 + (it - 1) * 6 + 0);
#pragma unroll
            for (int p = 0; p < PPT; p++)
#pragma unroll
                for (int c = 0; c < CPW; c++) {
                    io[p][c] = qv[p * CPW + c];
                    stash[(p * CPW + c) * kWG + tl] = io[p][c];
                }
            gauss();
#pragma unroll
            for (int p = 0; p < PPT; p++)
#pragma unroll
                for (int c = 0; c < CPW; c++) {
                    float v = pu[p][c];
                    const float m1 = (-a.wg) * io[p][c];
                    v = v - m1;
                    io[p][c] = stash[(p * CPW + c) * kWG + tl];
                    stash[(p * CPW + c) * kWG + tl] = v;
                }
            DSRG_PSTAMP(2 + (it - 1) * 6 + 1);
        };
        if (flags_g & 1) {
            // diagonal Gaussian lattice (training scale): per-pixel arithmetic, done inside the bilateral filter's index-load
            // shadow together with the wait for the marginals
            FilterIdx<VPT_B, PPT, 5> bix;
            filter_lattice<CPW, VPT_B, PPT, 5, true>(a.Lb, b, nullptr, nullptr, nc, N, val, inq, nullptr, io, tl, Mb, flags_b, [&]() {
                fetch_and_gauss([&]() {
                    FilterIdx<VPT_G, PPT, 2> gix;
                    filter_lattice<CPW, VPT_G, PPT, 2, true>(a.Lg, 0, nullptr, nullptr, nc, N, val, inq, nullptr, io, tl, Mg, flags_g,
                                                             NoMid(), 0, gix);
                });
            }, 0, bix);
        } else {
            fetch_and_gauss([&]() {
                FilterIdx<VPT_G, PPT, 2> gix;
                filter_lattice<CPW, VPT_G, PPT, 2, true>(a.Lg, 0, nullptr, nullptr, nc, N, val, inq, nullptr, io, tl, Mg, flags_g,
                                                         NoMid(), 0, gix);
            });
            __syncthreads();                                           // LDS is reused
            FilterIdx<VPT_B, PPT, 5> bix;
            filter_lattice<CPW, VPT_B, PPT, 5, true>(a.Lb, b, nullptr, nullptr, nc, N, val, inq, nullptr, io, tl, Mb, flags_b,
                                                     NoMid(), 0, bix);
        }
        DSRG_PSTAMP(2 + (it - 1) * 6 + 2);
#pragma unroll
        for (int p = 0; p < PPT; p++) {
            const int i = tid + p * kWG;
#pragma unroll
            for (int c = 0; c < CPW; c++) {
                if (i < N && c < nc) {
                    float v = stash[(p * CPW + c) * kWG + tid];
                    const float m2 = (-a.wb) * io[p][c];
                    v = v - m2;
                    store_granule(a.vg + img + (size_t)(c0 + c) * N + i, tag_v, v);
                }
            }
        }
        __syncthreads();
        DSRG_PSTAMP(2 + (it - 1) * 6 + 3);
        update_phase<true>(a, b, px0, px1, it, it == a.n_iters, scr);
        DSRG_PSTAMP(2 + (it - 1) * 6 + 4);
    }
#undef DSRG_PSTAMP
}

// ---------------------------------------------------------------------------------
void *g_filter_dbg = nullptr;   // set through dsrg_debug_set_filter_trace (tools only)

template <int CPW_B, int CPW_G, int VPT_B, int PPT>
static int launch_filter(const FilterArgs &a, int nblocks, size_t lds, hipStream_t stream, Profiler *prof) {
    static LdsGrant granted;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&mf_filter_kernel<CPW_B, CPW_G, VPT_B, PPT>), lds, granted);
    if (rc) return rc;
    const bool timed = prof && prof->active && prof->used < prof->cap;
    if (timed) DSRG_HIP_CHECK(hipEventRecord(prof->start[prof->used], stream));
    hipLaunchKernelGGL((mf_filter_kernel<CPW_B, CPW_G, VPT_B, PPT>), dim3(nblocks), dim3(kWG), lds, stream, a);
    DSRG_LAUNCH_CHECK();
    if (timed) { DSRG_HIP_CHECK(hipEventRecord(prof->stop[prof->used], stream)); prof->used++; }
    return DSRG_OK;
}

template <int CPW_B, int CPW_G>
static int dispatch_vpt(const FilterArgs &a, int nblocks, size_t lds, int vpt, hipStream_t stream, Profiler *prof) {
    if (vpt <= 4) return launch_filter<CPW_B, CPW_G, 4, 1>(a, nblocks, lds, stream, prof);
    if (vpt <= 10) return launch_filter<CPW_B, CPW_G, 10, 2>(a, nblocks, lds, stream, prof);
    if (vpt <= 16) return launch_filter<CPW_B, CPW_G, 16, 3>(a, nblocks, lds, stream, prof);
    if (vpt <= 25) return launch_filter<CPW_B, CPW_G, 25, 5>(a, nblocks, lds, stream, prof);
    if (vpt <= 32) return launch_filter<CPW_B, CPW_G, 32, 6>(a, nblocks, lds, stream, prof);
    return set_error(DSRG_ERR_UNSUPPORTED, "lattice too large for the LDS-resident filter (vpt=%d)", vpt);
}

template <int CPW, int VPT_B, int PPT>
static int launch_persistent(const PersistArgs &a, int nblocks, size_t lds, hipStream_t stream, Profiler *prof) {
    static LdsGrant granted;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&mf_persistent_kernel<CPW, VPT_B, PPT>), lds, granted);
    if (rc) return rc;
    const bool timed = prof && prof->active && prof->used < prof->cap;
    if (timed) DSRG_HIP_CHECK(hipEventRecord(prof->start[prof->used], stream));
    hipLaunchKernelGGL((mf_persistent_kernel<CPW, VPT_B, PPT>), dim3(nblocks), dim3(kWG), lds, stream, a);
    DSRG_LAUNCH_CHECK();
    if (timed) { DSRG_HIP_CHECK(hipEventRecord(prof->stop[prof->used], stream)); prof->used++; }
    return DSRG_OK;
}
// instantiated for maps up to 10 vertices per thread only (41x41 and smaller: the training sizes); larger LDS-resident maps
// (65x65) keep the launch-per-iteration loop
constexpr int kPersistMaxVpt = 10;
template <int CPW>
static int dispatch_persistent(const PersistArgs &a, int nblocks, size_t lds, int vpt, hipStream_t stream, Profiler *prof) {
    if (vpt <= 4) return launch_persistent<CPW, 4, 1>(a, nblocks, lds, stream, prof);
    if (vpt <= 10) return launch_persistent<CPW, 10, 2>(a, nblocks, lds, stream, prof);
    return set_error(DSRG_ERR_UNSUPPORTED, "lattice too large for the one-launch inference loop (vpt=%d)", vpt);
}

// Which inference loop runs: the launch-per-iteration loop (default; one filter + one update launch per iteration) or the
// one-launch loop (mf_persistent_kernel), selected by DSRG_MEANFIELD=persistent.  Measured on MI355X (profiles/
// r02_persistent_phase_trace.txt) the two take the same time at one image (0.446 ms per supervision step) and the one-launch
// loop is SLOWER at 16 images (0.61 vs 0.53 ms): a hand-off between the workgroups of an image costs two memory round trips
// (~3 us, what a kernel boundary plus the next kernel's first loads cost) and there are two per iteration, while the
// per-iteration index traffic that dominates either way (~400 KB per workgroup) cannot stay resident — registers and LDS
// are full.  Both produce bit-identical marginals (tests/test_gpu_parity.py).
int g_meanfield_mode = -1;     // -1: from the environment at first use; 0: launches; 1: persistent (dsrg_debug_set_meanfield_mode)
static bool persistent_enabled() {
    if (g_meanfield_mode < 0) {
        const char *e = getenv("DSRG_MEANFIELD");
        g_meanfield_mode = (e && strcmp(e, "persistent") == 0) ? 1 : 0;
    }
    return g_meanfield_mode != 0;
}

static int launch_update(const float *neg_unary, const MeanfieldBufs &buf, float wg, float wb, int use_msgs,
                         float *q_out, double *refined, float *logq, int B, int C, int N, hipStream_t stream) {
    // DSRG_UPDATE=pixel selects the one-thread-per-pixel kernel (round 1); default: kUpdParts threads per pixel
    static const bool split = [] { const char *e = getenv("DSRG_UPDATE"); return !(e && strcmp(e, "pixel") == 0); }();
    if (split) {
        const int blocks = (B * N + kUpdPix - 1) / kUpdPix;
#define DSRG_UPDS(CT_, UM_)                                                                                               \
    hipLaunchKernelGGL((mf_update_split_kernel<CT_, UM_>), dim3(blocks), dim3(kUpdParts * kUpdPix), 0, stream, neg_unary,     \
                       buf.msg_g, buf.msg_b, wg, wb, q_out, refined, logq, B, C, N, buf.work_counter)
        if (C <= 21) { if (use_msgs) DSRG_UPDS(21, true); else DSRG_UPDS(21, false); }
        else { if (use_msgs) DSRG_UPDS(kMaxLabels, true); else DSRG_UPDS(kMaxLabels, false); }
#undef DSRG_UPDS
        DSRG_LAUNCH_CHECK();
        return DSRG_OK;
    }
    const int threads = 256, blocks = (B * N + threads - 1) / threads;
#define DSRG_UPD(CT_, UM_)                                                                                  \
    hipLaunchKernelGGL((mf_update_kernel<CT_, UM_>), dim3(blocks), dim3(threads), 0, stream, neg_unary,        \
                       buf.msg_g, buf.msg_b, wg, wb, q_out, refined, logq, B, C, N, buf.work_counter)
    if (C <= 21) { if (use_msgs) DSRG_UPD(21, true); else DSRG_UPD(21, false); }
    else { if (use_msgs) DSRG_UPD(kMaxLabels, true); else DSRG_UPD(kMaxLabels, false); }
#undef DSRG_UPD
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

int launch_meanfield(const LatticeView &Lg, const LatticeView &Lb, const MeanfieldBufs &buf, int B, int C,
                     const float *neg_unary, float wg, float wb, int n_iters, float *q_out,
                     double *refined_out, float *logq_out, hipStream_t stream, Profiler *prof) {
    if (C < 1 || C > kMaxLabels) return set_error(DSRG_ERR_UNSUPPORTED, "1 <= nlabels <= %d required", kMaxLabels);
    const int N = Lb.N;
    const int vs_b = (Lb.Mcap + 1 + 3) & ~3, vs_g = (Lg.Mcap + 1 + 3) & ~3;
    const size_t kLds = 150 * 1024;
    auto lds_for = [&](int cpw_b, int cpw_g) {
        const size_t lb = (size_t)cpw_b * ((size_t)vs_b + N) * sizeof(float);
        const size_t lg = (size_t)cpw_g * ((size_t)vs_g + N) * sizeof(float);
        return lb > lg ? lb : lg;
    };
    // planes per block: bilateral 2 (8-byte LDS gathers) and Gaussian 4 (16-byte) when LDS holds them and
    // the batch is large enough to still fill the chip; otherwise narrower blocks
    int cpw_b = 1, cpw_g = 1;
    if (lds_for(2, 4) <= kLds && (size_t)B * ((C + 1) / 2) >= 64) { cpw_b = 2; cpw_g = 4; }
    else if (lds_for(1, 2) <= kLds) { cpw_b = 1; cpw_g = 2; }
    else if (lds_for(1, 1) > 158 * 1024) return set_error(DSRG_ERR_UNSUPPORTED, "lattice does not fit LDS");
    const int vpt = (Lb.Mcap + kWG - 1) / kWG;
    const int ppt_tab = vpt <= 4 ? 1 : vpt <= 10 ? 2 : vpt <= 16 ? 3 : vpt <= 25 ? 5 : 6;
    if (N > ppt_tab * kWG) return set_error(DSRG_ERR_UNSUPPORTED, "pixel count %d exceeds the filter kernel", N);
    if (Lg.Mcap > ((vpt <= 4 ? 4 : vpt <= 10 ? 10 : vpt <= 16 ? 16 : vpt <= 25 ? 25 : 32) + 1) / 2 * kWG)
        return set_error(DSRG_ERR_UNSUPPORTED, "Gaussian lattice exceeds the filter kernel");

    // ---- one launch for the whole loop (mf_persistent_kernel) whenever the hand-off buffers exist
    if (persistent_enabled() && buf.qg && buf.vg && buf.status && buf.epoch && n_iters >= 1 && n_iters <= 62 &&
        vpt <= kPersistMaxVpt) {
        const int ppt_p = vpt <= 4 ? 1 : vpt <= 10 ? 2 : vpt <= 16 ? 3 : vpt <= 25 ? 5 : 6;
        auto lds_p = [&](int cpw) {        // lattice values + normalised input planes + scratch + parked per-thread values
            return (size_t)cpw * ((size_t)vs_b + N) * sizeof(float) + (size_t)(1 + ppt_p * cpw) * kWG * sizeof(float);
        };
        // one label plane per workgroup while that still fits one round of 256 CUs (a 1-plane block is no slower than a
        // 2-plane one is faster: the same number of LDS gathers), two planes (8-byte gathers) beyond
        int cpw = ((size_t)B * C <= 256 || lds_p(2) > kLds) ? 1 : 2;
        if (lds_p(cpw) <= kLds) {
            PersistArgs p;
            p.Lg = Lg; p.Lb = Lb; p.neg_unary = neg_unary; p.qg = buf.qg; p.vg = buf.vg;
            p.q_out = q_out; p.refined_out = refined_out; p.logq_out = logq_out; p.status = buf.status;
            p.wg = wg; p.wb = wb; p.C = C; p.N = N; p.n_iters = n_iters;
            p.groups = (C + cpw - 1) / cpw;
            p.lds_val_stride = vs_b;
            p.lp_shift = C <= 32 ? 5 : 6;
            const int vpt_p = (Lb.Mcap + kWG - 1) / kWG;
            p.dbg = reinterpret_cast<unsigned long long *>(g_filter_dbg);
            // Every workgroup of an image must be resident at once, and one workgroup (16 waves, > 64 VGPRs) fills a CU:
            // at most 256 per launch, and — the dispatcher deals blocks to the 8 XCDs round-robin — at most 32 per XCD.
            const int bmax = 256 / p.groups > 0 ? 256 / p.groups : 1;
            for (int b0 = 0; b0 < B; b0 += bmax) {
                const int Bc = B - b0 < bmax ? B - b0 : bmax;
                p.b0 = b0; p.B = b0 + Bc;
                // preferred map: the workgroups of an image share blockIdx % 8 (one XCD: its L2 holds the image's index
                // arrays) for the largest multiple of 8 images, the rest image-major; taken only if no XCD gets more than 32
                p.lat_stride = Bc < 8 ? 8 : (Bc & ~7);
                p.nblk_xcd = p.groups * p.lat_stride;
                int tail = Bc > p.lat_stride ? p.groups * (Bc - p.lat_stride) : 0;
                int worst = 0;
                for (int x = 0; x < 8; x++) {
                    int load = 0;
                    for (int bb = x; bb < Bc && bb < p.lat_stride; bb += 8) load += p.groups;
                    load += (tail + 7 - x) / 8;
                    worst = load > worst ? load : worst;
                }
                if (worst > 32) {                  // spread every image over the XCDs instead (image-major everywhere)
                    p.lat_stride = 0; p.nblk_xcd = 0; tail = p.groups * Bc;
                }
                const int nblk = p.nblk_xcd + tail;
                unsigned int e = ++*buf.epoch;
                if ((e & 0x03FFFFFFu) == 0) e = ++*buf.epoch;                  // tag 0 is the "never written" state
                p.epoch = e & 0x03FFFFFFu;
                int rc = cpw == 2 ? dispatch_persistent<2>(p, nblk, lds_p(2), vpt_p, stream, prof)
                                  : dispatch_persistent<1>(p, nblk, lds_p(1), vpt_p, stream, prof);
                if (rc) return rc;
            }
            return DSRG_OK;
        }
    }

    FilterArgs a;
    a.Lg = Lg; a.Lb = Lb; a.q = buf.q; a.msg_g = buf.msg_g; a.msg_b = buf.msg_b;
    a.B = B; a.C = C; a.N = N;
    a.groups_b = (C + cpw_b - 1) / cpw_b;
    // one workgroup per CU (LDS), 32 CUs per XCD, blockIdx % 8 picks the XCD: keep whole images on one XCD for the largest
    // multiple of 8 images (small batches are padded up to 8), spread the remaining images' blocks over all XCDs
    a.lat_stride = B < 8 ? 8 : (B & ~7);
    a.nblk_xcd = a.groups_b * a.lat_stride;
    a.nblk_b = a.nblk_xcd + (B > a.lat_stride ? a.groups_b * (B - a.lat_stride) : 0);
    // Gaussian blocks come last in the grid and fill CUs as bilateral blocks (twice as long) drain; one image each
    // balances best up to B = 24, two images each beyond (measured: B = 16 20.0 vs 20.8 us, B = 32 39.2 vs 38.1 us)
    // Gaussian units come last in the unit list: the CUs the bilateral units leave idle start on them at once and the rest
    // are pulled from the counter as workgroups finish (see mf_filter_kernel)
    a.groups_g = (C + cpw_g - 1) / cpw_g;
    a.nunits = a.nblk_b + a.groups_g * B;
    a.counter = buf.work_counter;
    a.lds_val_stride = vs_b;
    a.lds_val_stride_g = vs_g;
    a.dbg = reinterpret_cast<unsigned long long *>(g_filter_dbg);
    static const int n_cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    // one workgroup per CU (its LDS); without a counter (or with DSRG_FILTER_QUEUE=0) every unit gets its own workgroup and
    // the hardware dispatcher hands them out
    // (measured, profiles/r02_filter_queue_ab.txt: the software queue — DSRG_FILTER_QUEUE=1 — loses to the hardware dispatcher
    // at every batch size: 19.0-19.7 vs 18.6-19.3 us per launch at 16 images, 29.8-30.1 vs 27.1-27.6 at 20, 39.8-40.2 vs
    // 36.0-36.5 at 32; it stays selectable for that comparison)
    static const bool use_queue = [] { const char *e = getenv("DSRG_FILTER_QUEUE"); return e && e[0] == '1'; }();
    const int nblocks = (use_queue && buf.work_counter && a.nunits > n_cus) ? n_cus : a.nunits;
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
    lds += 16;                                       // the next-unit broadcast slot behind the region

    // Q0 = expAndNormalize(-unary)   (densecrf.cpp:120)
    int rc = launch_update(neg_unary, buf, wg, wb, 0, n_iters > 0 ? buf.q : q_out,
                           n_iters > 0 ? nullptr : refined_out, n_iters > 0 ? nullptr : logq_out, B, C, N, stream);
    if (rc) return rc;
    for (int it = 0; it < n_iters; it++) {
        if (cpw_b == 2) rc = dispatch_vpt<2, 4>(a, nblocks, lds, vpt, stream, prof);
        else if (cpw_g == 2) rc = dispatch_vpt<1, 2>(a, nblocks, lds, vpt, stream, prof);
        else rc = dispatch_vpt<1, 1>(a, nblocks, lds, vpt, stream, prof);
        if (rc) return rc;
        const bool last = (it == n_iters - 1);
        rc = launch_update(neg_unary, buf, wg, wb, 1, last ? q_out : buf.q, last ? refined_out : nullptr,
                           last ? logq_out : nullptr, B, C, N, stream);
        if (rc) return rc;
    }
    return DSRG_OK;
}

}  // namespace dsrg
