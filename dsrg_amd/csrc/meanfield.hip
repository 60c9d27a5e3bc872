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
    int gau_stride;          // Gaussian blocks per plane group: ceil(B / ipb)
    int ipb;                 // images per Gaussian block
    int lds_val_stride;      // bilateral: vertices in the LDS value array (Mcap_b + 1, padded to 4)
    int lds_val_stride_g;    // Gaussian:  (Mcap_g + 1, padded to 4)
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
template <int CPW, int VPT, int PPT, int D>
__device__ __forceinline__ void filter_lattice(const LatticeView &L, int li, const float *__restrict__ qb,
                                               float *__restrict__ out, int nc, int N,
                                               typename PlaneVec<CPW>::type *val,
                                               typename PlaneVec<CPW>::type *inq, unsigned long long *dbg) {
    using vec_t = typename PlaneVec<CPW>::type;
    constexpr int D1 = D + 1;
    constexpr bool DEEP = VPT <= 10;             // all index words of a thread fit the register file
    constexpr int KC = DEEP ? VPT : 8;           // vertices per chunk of index loads otherwise
    constexpr int NCH = (VPT + KC - 1) / KC;
    const int tid = threadIdx.x;
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

    const int M = L.M[li];
    DSRG_STAMP(0);

    if (L.flags[li] & 1) {
        // Diagonal lattice (e.g. the Gaussian kernel at training scale: sigma = 0.25 px):
        // every simplex corner is private to its pixel and has no blur neighbour, so splat, blur and
        // slice collapse to per-pixel arithmetic — evaluated here in the general path's operation
        // order (products, 0 + p, val + 0.5*(0+0), ordered slice sum), hence bit-identical to it.
        const float alpha = 1.0f / (1.0f + exp2f(-(float)D));
#pragma unroll
        for (int p = 0; p < PPT; p++) {
            const int i = tid + p * kWG;
            const float nv = ld_f32(r_norm, (uint32_t)tid * 4u, (uint32_t)p * (kWG * 4u));
            float bw[D1], qq[CPW];
#pragma unroll
            for (int r = 0; r < D1; r++)
                bw[r] = ld_f32(r_bary, (uint32_t)tid * 4u, ((uint32_t)p * kWG + (uint32_t)r * (uint32_t)N) * 4u);
#pragma unroll
            for (int c = 0; c < CPW; c++)
                qq[c] = ld_f32(r_q, (uint32_t)tid * 4u, (uint32_t)p * (kWG * 4u) + (uint32_t)c * (uint32_t)N * 4u);
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
                        out[(size_t)c * N + i] = acc * nv;
                    }
                }
            }
        }
        DSRG_STAMP(11);
        if (dbg && tid == 0) dbg[12] = (unsigned long long)M;
        return;
    }

    // ---- stage A: everything that does not depend on anything, in one burst
    float qv[PPT][CPW], nrm[PPT];
#pragma unroll
    for (int p = 0; p < PPT; p++) {
        nrm[p] = ld_f32(r_norm, (uint32_t)tid * 4u, (uint32_t)p * (kWG * 4u));
#pragma unroll
        for (int c = 0; c < CPW; c++)
            qv[p][c] = ld_f32(r_q, (uint32_t)tid * 4u, (uint32_t)p * (kWG * 4u) + (uint32_t)c * (uint32_t)N * 4u);
    }
    uint32_t epx[KC];
    float ew[KC];
    uint32_t rs0[KC], rs1[KC];
#pragma unroll
    for (int k = 0; k < KC; k++) {                 // splat entries e_k = tid + k*1024 and CSR rows of v_k
        epx[k] = ld_u16(r_cp, (uint32_t)tid * 2u, (uint32_t)k * (kWG * 2u));
        ew[k] = ld_f32(r_cw, (uint32_t)tid * 4u, (uint32_t)k * (kWG * 4u));
        rs0[k] = ld_u32(r_rs, (uint32_t)tid * 4u, (uint32_t)k * (kWG * 4u));
        rs1[k] = ld_u32(r_rs, (uint32_t)tid * 4u, (uint32_t)k * (kWG * 4u) + 4u);
    }
    // neighbour words n1 | n2<<16 of my vertices: a ring of RING axes, fetched RING-1 passes ahead of
    // their use (a blur pass is shorter than one memory round trip)
    constexpr int RING = 3;
    uint32_t nbw[DEEP ? RING : 1][KC];
    auto load_axis = [&](int j) {
#pragma unroll
        for (int k = 0; k < KC; k++)
            nbw[j % RING][k] = ld_u32(r_nb, (uint32_t)tid * 4u, (uint32_t)j * (uint32_t)Mcap * 4u + (uint32_t)k * (kWG * 4u));
    };
    if constexpr (DEEP) load_axis(0);

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
    vec_t sacc[VPT];
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
    __syncthreads();                                     // every row of products has been consumed
#pragma unroll
    for (int k = 0; k < VPT; k++) {
        const int v = tid + k * kWG;
        if (v < M) val[v] = sacc[k];
    }
    if (tid == 0) {                                     // zero sentinel = "no neighbour" (permutohedral.cpp:561-562)
        vec_t z;
#pragma unroll
        for (int c = 0; c < CPW; c++) pv_set(z, c, 0.0f);
        val[Mcap] = z;
    }
    __syncthreads();
    DSRG_STAMP(4);

    // ---- blur along the d+1 lattice axes (permutohedral.cpp:556-569): Jacobi per axis — new values
    // held in registers between the read barrier and the write barrier
    uint32_t sv[PPT][D1];
    float sw[PPT][D1];
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
                    const uint32_t word = nbw[DEEP ? j % RING : 0][k];
                    const bool ok = tid + (ch * KC + k) * kWG < M;
                    const int n1 = ok ? (int)(word & 0xffffu) : Mcap, n2 = ok ? (int)(word >> 16) : Mcap;
                    const vec_t x1 = val[n1], x2 = val[n2];
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
        if (j == (D1 >= 3 ? D1 - 3 : 0)) {           // slice corners, two passes ahead of their use
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
        __syncthreads();                             // all gathers of this axis are done
#pragma unroll
        for (int k = 0; k < VPT; k++) {
            const int v = tid + k * kWG;
            if (v < M) val[v] = sacc[k];
        }
        __syncthreads();
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
                const vec_t x = val[min(sv[p][r], (uint32_t)Mcap)];
#pragma unroll
                for (int c = 0; c < CPW; c++) acc[c] = acc[c] + w * pv_get(x, c);
            }
#pragma unroll
            for (int c = 0; c < CPW; c++)
                if (c < nc) out[(size_t)c * N + i] = acc[c] * nrm[p];
        }
    }
    DSRG_STAMP(11);
    if (dbg && tid == 0) dbg[12] = (unsigned long long)M;
}

// One launch filters every label plane of every image through both lattices.  The bilateral work
// (6 axes, M ~ 2-6 N vertices per image) goes to blocks of CPW_B planes of one image; the Gaussian
// work (3 axes, one lattice shared by all images) to blocks of CPW_G planes of IPB images — the
// block counts are chosen by the host so that both kinds finish together and all fit one round.
template <int CPW_B, int CPW_G, int VPT_B, int PPT>
__global__ __launch_bounds__(kWG) void mf_filter_kernel(FilterArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int VPT_G = (VPT_B + 1) / 2;                 // Mcap_gauss = Mcap_bilateral / 2
    unsigned long long *dbg = a.dbg ? a.dbg + (size_t)blockIdx.x * 32 : nullptr;
    if ((int)blockIdx.x < a.nblk_b) {
        using vec_t = typename PlaneVec<CPW_B>::type;
        // blocks of one image share blockIdx % 8, i.e. one XCD and its L2; the images beyond the last multiple of 8
        // (B = 20: images 16..19) are laid out image-major so that they spread evenly over the XCDs — with the
        // XCD-aware order alone 3 images x 11 blocks would land on a 32-CU XCD and force a second round there
        int g, b;
        if ((int)blockIdx.x < a.nblk_xcd) { g = blockIdx.x / a.lat_stride; b = blockIdx.x % a.lat_stride; }
        else { const int rel = blockIdx.x - a.nblk_xcd; b = a.lat_stride + rel / a.groups_b; g = rel % a.groups_b; }
        if (b >= a.B) return;
        const int c0 = g * CPW_B, nc = min(CPW_B, a.C - c0);
        vec_t *val = reinterpret_cast<vec_t *>(smem);                      // [Mcap + 1] label-interleaved
        vec_t *inq = val + a.lds_val_stride;                               // [N]
        const size_t o = ((size_t)b * a.C + c0) * a.N;
        filter_lattice<CPW_B, VPT_B, PPT, 5>(a.Lb, b, a.q + o, a.msg_b + o, nc, a.N, val, inq, dbg);
    } else {
        using vec_t = typename PlaneVec<CPW_G>::type;
        const int rel = blockIdx.x - a.nblk_b;
        const int g = rel / a.gau_stride, pb = rel % a.gau_stride;
        const int c0 = g * CPW_G, nc = min(CPW_G, a.C - c0);
        vec_t *val = reinterpret_cast<vec_t *>(smem);
        vec_t *inq = val + a.lds_val_stride_g;
        for (int ii = 0; ii < a.ipb; ii++) {
            const int b = pb * a.ipb + ii;
            if (b >= a.B) break;
            if (ii) __syncthreads();                                       // LDS is reused
            const size_t o = ((size_t)b * a.C + c0) * a.N;
            filter_lattice<CPW_G, VPT_G, PPT, 2>(a.Lg, 0, a.q + o, a.msg_g + o, nc, a.N, val, inq,
                                                 (dbg && ii == 0) ? dbg + 16 : nullptr);
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
                                                        float *__restrict__ logq_out, int B, int C, int N) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
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

// ---------------------------------------------------------------------------------
void *g_filter_dbg = nullptr;   // set through dsrg_debug_set_filter_trace (tools only)

template <int CPW_B, int CPW_G, int VPT_B, int PPT>
static int launch_filter(const FilterArgs &a, int nblocks, size_t lds, hipStream_t stream, Profiler *prof) {
    static size_t granted = 0;
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

static int launch_update(const float *neg_unary, const MeanfieldBufs &buf, float wg, float wb, int use_msgs,
                         float *q_out, double *refined, float *logq, int B, int C, int N, hipStream_t stream) {
    const int threads = 256, blocks = (B * N + threads - 1) / threads;
#define DSRG_UPD(CT_, UM_)                                                                                  \
    hipLaunchKernelGGL((mf_update_kernel<CT_, UM_>), dim3(blocks), dim3(threads), 0, stream, neg_unary,        \
                       buf.msg_g, buf.msg_b, wg, wb, q_out, refined, logq, B, C, N)
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
    const int groups_g = (C + cpw_g - 1) / cpw_g;
    a.ipb = B > 24 ? 2 : 1;
    a.gau_stride = (B + a.ipb - 1) / a.ipb;
    a.lds_val_stride = vs_b;
    a.lds_val_stride_g = vs_g;
    a.dbg = reinterpret_cast<unsigned long long *>(g_filter_dbg);
    const int nblocks = a.nblk_b + groups_g * a.gau_stride;
    const size_t lds = lds_for(cpw_b, cpw_g);

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
