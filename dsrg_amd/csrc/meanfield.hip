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
    int groups;              // ceil(C / CPW)
    int lat_stride;          // round_up(2B, 8)
    int lds_val_stride;      // floats per label plane of lattice values (Mcap_max + 1, padded)
};

template <int CPW, int VPT>
__global__ __launch_bounds__(kWG) void mf_filter_kernel(FilterArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int g = blockIdx.x / a.lat_stride, lat = blockIdx.x % a.lat_stride;
    if (lat >= 2 * a.B) return;
    const int b = lat >> 1, kind = lat & 1;      // kind 0 = Gaussian, 1 = bilateral
    const LatticeView &L = kind ? a.Lb : a.Lg;
    const int li = kind ? b : 0;                 // lattice index inside its set
    const int N = a.N, D1 = L.d + 1, Mcap = L.Mcap;
    const int M = L.M[li];
    const int c0 = g * CPW;
    const int nc = min(CPW, a.C - c0);

    float *val = reinterpret_cast<float *>(smem);                          // [CPW][lds_val_stride]
    float *inq = val + (size_t)CPW * a.lds_val_stride;                     // [CPW][N]
    const int VS = a.lds_val_stride;

    const uint16_t *vid = L.vid + (size_t)li * D1 * N;
    const float *bary = L.bary + (size_t)li * D1 * N;
    const uint32_t *nb = L.nb + (size_t)li * D1 * Mcap;
    const uint32_t *row_start = L.row_start + (size_t)li * (Mcap + 1);
    const uint16_t *csr_pix = L.csr_pix + (size_t)li * D1 * N;
    const float *csr_w = L.csr_w + (size_t)li * D1 * N;
    const float *norm = L.norm + (size_t)li * N;
    const float *qb = a.q + ((size_t)b * a.C + c0) * N;
    float *out = (kind ? a.msg_b : a.msg_g) + ((size_t)b * a.C + c0) * N;

    // in = Q * norm   (pairwise.cpp:66)
    for (int i = tid; i < N; i += kWG) {
        const float nv = norm[i];
#pragma unroll
        for (int c = 0; c < CPW; c++) inq[c * N + i] = (c < nc) ? qb[(size_t)c * N + i] * nv : 0.0f;
    }
    if (tid < CPW) val[tid * VS + Mcap] = 0.0f;          // zero sentinel = "no neighbour" (permutohedral.cpp:561-562)
    __syncthreads();

    // splat (permutohedral.cpp:545-553) as a per-vertex gather in the reference's accumulation order
    for (int v = tid; v < M; v += kWG) {
        float s[CPW];
#pragma unroll
        for (int c = 0; c < CPW; c++) s[c] = 0.0f;
        const uint32_t p0 = row_start[v], p1 = row_start[v + 1];
        for (uint32_t pos = p0; pos < p1; pos++) {
            const int px = csr_pix[pos];
            const float w = csr_w[pos];
#pragma unroll
            for (int c = 0; c < CPW; c++) s[c] = s[c] + w * inq[c * N + px];
        }
#pragma unroll
        for (int c = 0; c < CPW; c++) val[c * VS + v] = s[c];
    }
    __syncthreads();

    // blur along the d+1 lattice axes (permutohedral.cpp:556-569): Jacobi per axis —
    // new values held in registers between the read barrier and the write barrier
    for (int j = 0; j < D1; j++) {
        float nv[CPW][VPT];
        const uint32_t *nbj = nb + (size_t)j * Mcap;
#pragma unroll
        for (int k = 0; k < VPT; k++) {
            const int v = tid + k * kWG;
            if (v < M) {
                const uint32_t t = nbj[v];
                const int n1 = t & 0xffffu, n2 = t >> 16;
#pragma unroll
                for (int c = 0; c < CPW; c++) {
                    float s = val[c * VS + n1] + val[c * VS + n2];
                    s = 0.5f * s;
                    nv[c][k] = val[c * VS + v] + s;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < VPT; k++) {
            const int v = tid + k * kWG;
            if (v < M) {
#pragma unroll
                for (int c = 0; c < CPW; c++) val[c * VS + v] = nv[c][k];
            }
        }
        __syncthreads();
    }

    // slice (permutohedral.cpp:571-584), then out * norm (pairwise.cpp:79)
    const float alpha = 1.0f / (1.0f + exp2f(-(float)L.d));
    for (int i = tid; i < N; i += kWG) {
        float acc[CPW];
#pragma unroll
        for (int c = 0; c < CPW; c++) acc[c] = 0.0f;
        for (int r = 0; r < D1; r++) {
            const int v = vid[(size_t)r * N + i];
            const float w = bary[(size_t)r * N + i] * alpha;
#pragma unroll
            for (int c = 0; c < CPW; c++) acc[c] = acc[c] + w * val[c * VS + v];
        }
        const float nv = norm[i];
#pragma unroll
        for (int c = 0; c < CPW; c++)
            if (c < nc) out[(size_t)c * N + i] = acc[c] * nv;
    }
}

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

template <int CT>   // CT = compile-time bound on C (loops fully unrolled, values in registers)
__global__ __launch_bounds__(256) void mf_update_kernel(const float *__restrict__ neg_unary,
                                                        const float *__restrict__ msg_g,
                                                        const float *__restrict__ msg_b, float wg, float wb,
                                                        int use_msgs, float *__restrict__ q_out,
                                                        double *__restrict__ refined_out,
                                                        float *__restrict__ logq_out, int B, int C, int N) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * N) return;
    const int b = idx / N, i = idx - b * N;
    const size_t base = (size_t)b * C * N + i;
    float t[CT];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < CT; c++) {
        if (c < C) {
            float v = neg_unary[base + (size_t)c * N];               // tmp1 = -unary
            if (use_msgs) {
                // tmp2 = -w * filter(Q); tmp1 -= tmp2  — Gaussian first, then bilateral
                float m1 = (-wg) * msg_g[base + (size_t)c * N];
                v = v - m1;
                float m2 = (-wb) * msg_b[base + (size_t)c * N];
                v = v - m2;
            }
            t[c] = v;
            mx = fmaxf(mx, v);
        }
    }
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < CT; c++) {
        if (c < C) { t[c] = expf(t[c] - mx); sum = sum + t[c]; }
    }
#pragma unroll
    for (int c = 0; c < CT; c++) {
        if (c < C) t[c] = t[c] / sum;
    }
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
template <int CPW, int VPT>
static int launch_filter(const FilterArgs &a, size_t lds, hipStream_t stream, Profiler *prof) {
    static size_t granted = 0;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&mf_filter_kernel<CPW, VPT>), lds, granted);
    if (rc) return rc;
    const bool timed = prof && prof->active && prof->used < prof->cap;
    if (timed) DSRG_HIP_CHECK(hipEventRecord(prof->start[prof->used], stream));
    hipLaunchKernelGGL((mf_filter_kernel<CPW, VPT>), dim3(a.groups * a.lat_stride), dim3(kWG), lds, stream, a);
    DSRG_LAUNCH_CHECK();
    if (timed) { DSRG_HIP_CHECK(hipEventRecord(prof->stop[prof->used], stream)); prof->used++; }
    return DSRG_OK;
}

template <int CPW>
static int dispatch_vpt(const FilterArgs &a, size_t lds, int vpt, hipStream_t stream, Profiler *prof) {
    if (vpt <= 4) return launch_filter<CPW, 4>(a, lds, stream, prof);
    if (vpt <= 10) return launch_filter<CPW, 10>(a, lds, stream, prof);
    if (vpt <= 16) return launch_filter<CPW, 16>(a, lds, stream, prof);
    if (vpt <= 32) return launch_filter<CPW, 32>(a, lds, stream, prof);
    return set_error(DSRG_ERR_UNSUPPORTED, "lattice too large for the LDS-resident filter (vpt=%d)", vpt);
}

static int launch_update(const float *neg_unary, const MeanfieldBufs &buf, float wg, float wb, int use_msgs,
                         float *q_out, double *refined, float *logq, int B, int C, int N, hipStream_t stream) {
    const int threads = 256, blocks = (B * N + threads - 1) / threads;
    if (C <= 21)
        hipLaunchKernelGGL(mf_update_kernel<21>, dim3(blocks), dim3(threads), 0, stream, neg_unary, buf.msg_g,
                           buf.msg_b, wg, wb, use_msgs, q_out, refined, logq, B, C, N);
    else
        hipLaunchKernelGGL(mf_update_kernel<kMaxLabels>, dim3(blocks), dim3(threads), 0, stream, neg_unary,
                           buf.msg_g, buf.msg_b, wg, wb, use_msgs, q_out, refined, logq, B, C, N);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

int launch_meanfield(const LatticeView &Lg, const LatticeView &Lb, const MeanfieldBufs &buf, int B, int C,
                     const float *neg_unary, float wg, float wb, int n_iters, float *q_out,
                     double *refined_out, float *logq_out, hipStream_t stream, Profiler *prof) {
    if (C < 1 || C > kMaxLabels) return set_error(DSRG_ERR_UNSUPPORTED, "1 <= nlabels <= %d required", kMaxLabels);
    const int N = Lb.N;
    const int McapMax = Lb.Mcap > Lg.Mcap ? Lb.Mcap : Lg.Mcap;
    const int vs = (McapMax + 1 + 3) & ~3;
    // label planes per workgroup: as many as LDS holds, but keep >= ~200 workgroups in flight
    const size_t per_plane = ((size_t)vs + (size_t)N) * sizeof(float);
    int cpw = 1;
    if (3 * per_plane <= 150 * 1024 && (size_t)2 * B * ((C + 2) / 3) >= 192) cpw = 3;
    else if (2 * per_plane <= 150 * 1024 && (size_t)2 * B * ((C + 1) / 2) >= 192) cpw = 2;
    if (per_plane > 158 * 1024) return set_error(DSRG_ERR_UNSUPPORTED, "lattice does not fit LDS");
    const int vpt = (McapMax + kWG - 1) / kWG;

    FilterArgs a;
    a.Lg = Lg; a.Lb = Lb; a.q = buf.q; a.msg_g = buf.msg_g; a.msg_b = buf.msg_b;
    a.B = B; a.C = C; a.N = N;
    a.groups = (C + cpw - 1) / cpw;
    a.lat_stride = (2 * B + 7) & ~7;
    a.lds_val_stride = vs;
    const size_t lds = (size_t)cpw * per_plane;

    // Q0 = expAndNormalize(-unary)   (densecrf.cpp:120)
    int rc = launch_update(neg_unary, buf, wg, wb, 0, n_iters > 0 ? buf.q : q_out,
                           n_iters > 0 ? nullptr : refined_out, n_iters > 0 ? nullptr : logq_out, B, C, N, stream);
    if (rc) return rc;
    for (int it = 0; it < n_iters; it++) {
        if (cpw == 3) rc = dispatch_vpt<3>(a, lds, vpt, stream, prof);
        else if (cpw == 2) rc = dispatch_vpt<2>(a, lds, vpt, stream, prof);
        else rc = dispatch_vpt<1>(a, lds, vpt, stream, prof);
        if (rc) return rc;
        const bool last = (it == n_iters - 1);
        rc = launch_update(neg_unary, buf, wg, wb, 1, last ? q_out : buf.q, last ? refined_out : nullptr,
                           last ? logq_out : nullptr, B, C, N, stream);
        if (rc) return rc;
    }
    return DSRG_OK;
}

}  // namespace dsrg
