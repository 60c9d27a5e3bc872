// Pointwise layers and input preparation of the DSRG supervision path (gfx950).
// NCHW float32 blobs; lanes map to pixels so every label plane is read coalesced.
//
// Replaces SoftmaxLayer, CRFLayer's pre/post-processing and backward,
// BalancedSeedLossLayer and ConstrainLossLayer of pylayers/pylayers/pylayers.py
// (line numbers on each kernel).  The Theano graphs are restated in closed form.
#include <math.h>
#include "common.h"

namespace dsrg {

// probs[probs < min_prob] = min_prob, in place   (pylayers.py:67,312)
__global__ void clip_min_kernel(float *__restrict__ p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float v = p[i];
        if (v < kMinProb) p[i] = kMinProb;
    }
}
int launch_clip_min(float *p, size_t n, hipStream_t stream) {
    const int threads = 256;
    const int blocks = (int)((n + threads - 1) / threads < 2048 ? (n + threads - 1) / threads : 2048);
    hipLaunchKernelGGL(clip_min_kernel, dim3(blocks > 0 ? blocks : 1), dim3(threads), 0, stream, p, n);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

// ---- SoftmaxLayer (pylayers.py:30-51) ----------------------------------------------------
// forward: s = softmax_c(x); p = (s + 1e-4) / sum_c(s + 1e-4), fp32
// Four adjacent lanes share a pixel and split its labels (c = part, part + 4, ...): the fp64-rounded exps of a pixel no
// longer queue up in one thread; the sums over the labels are formed in label order from shuffled terms (bit-identical
// to one thread per pixel).
constexpr int kSmParts = 4, kSmPix = 64;       // (8 lanes per pixel: measured 2-3 us per step slower at 16 images, equal for one)
template <int CT>
__global__ __launch_bounds__(kSmParts * kSmPix) void softmax_fwd_kernel(int B, int C, int HW, const float *__restrict__ x,
                                                                         float *__restrict__ p, float floor_at,
                                                                         float *__restrict__ q0) {
    constexpr int P = kSmParts, LPP = (CT + P - 1) / P;
    const int lane = threadIdx.x & 63, part = threadIdx.x & (P - 1), lane0 = lane & ~(P - 1);
    const int idx = blockIdx.x * kSmPix + (int)(threadIdx.x / P);
    const bool live = idx < B * HW;
    const int b = live ? idx / HW : 0, i = live ? idx - b * HW : 0;
    const size_t base = (size_t)b * C * HW + i;
    float t[LPP];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < LPP; k++) t[k] = x[base + (size_t)min(part + k * P, C - 1) * HW];    // unconditional: all in flight
#pragma unroll
    for (int k = 0; k < LPP; k++) { t[k] = (part + k * P < C) ? t[k] : -INFINITY; mx = fmaxf(mx, t[k]); }
#pragma unroll
    for (int m = 1; m < P; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
#pragma unroll
    for (int k = 0; k < LPP; k++) t[k] = exp_cr(t[k] - mx);
    float z = 0.0f;
#pragma unroll
    for (int k = 0; k < LPP; k++)
#pragma unroll
        for (int q = 0; q < P; q++) {
            const float e = __shfl(t[k], lane0 + q, 64);
            if (k * P + q < C) z = z + e;
        }
    float z2 = 0.0f;
#pragma unroll
    for (int k = 0; k < LPP; k++) t[k] = t[k] / z + kMinProb;
#pragma unroll
    for (int k = 0; k < LPP; k++)
#pragma unroll
        for (int q = 0; q < P; q++) {
            const float e = __shfl(t[k], lane0 + q, 64);
            if (k * P + q < C) z2 = z2 + e;
        }
    // floor_at = 0: the plain layer; 1e-4: the in-place clip CRFLayer.forward applies next (pylayers.py:67)
#pragma unroll
    for (int k = 0; k < LPP; k++) t[k] = fmaxf(t[k] / z2, floor_at);
#pragma unroll
    for (int k = 0; k < LPP; k++)
        if (live && part + k * P < C) p[base + (size_t)(part + k * P) * HW] = t[k];
    if (q0) {
        // the mean field's starting point Q0 = expAndNormalize(-unary) (densecrf.cpp:120; the unary energy is minus this blob,
        // CRF.py:28) in mf_update_split_kernel's arithmetic: column maximum, fp64-rounded exp, label-order sum, division
        float m2 = -INFINITY;
#pragma unroll
        for (int k = 0; k < LPP; k++) m2 = fmaxf(m2, (part + k * P < C) ? t[k] : -INFINITY);
#pragma unroll
        for (int m = 1; m < P; m <<= 1) m2 = fmaxf(m2, __shfl_xor(m2, m, 64));
#pragma unroll
        for (int k = 0; k < LPP; k++) t[k] = exp_cr(((part + k * P < C) ? t[k] : -INFINITY) - m2);
        float s2 = 0.0f;
#pragma unroll
        for (int k = 0; k < LPP; k++)
#pragma unroll
            for (int q = 0; q < P; q++) {
                const float e = __shfl(t[k], lane0 + q, 64);
                if (k * P + q < C) s2 = s2 + e;
            }
#pragma unroll
        for (int k = 0; k < LPP; k++)
            if (live && part + k * P < C) q0[base + (size_t)(part + k * P) * HW] = t[k] / s2;
    }
}
// backward = T.grad(sum(probs*g), preds):  dx_j = s_j (g_j - sum_k s_k g_k) / Z,  Z = sum_c (s_c + 1e-4)
template <int CT>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(int B, int C, int HW, const float *__restrict__ x,
                                                          const float *__restrict__ g, float *__restrict__ dx) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * HW) return;
    const int b = idx / HW, i = idx - b * HW;
    const size_t base = (size_t)b * C * HW + i;
    float s[CT], gg[CT];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < CT; c++) {
        const size_t o = base + (size_t)min(c, C - 1) * HW;
        s[c] = x[o];
        gg[c] = g[o];
    }
#pragma unroll
    for (int c = 0; c < CT; c++) { s[c] = (c < C) ? s[c] : -INFINITY; mx = fmaxf(mx, s[c]); }
    float z = 0.0f;
#pragma unroll
    for (int c = 0; c < CT; c++) { s[c] = exp_cr(s[c] - mx); z = (c < C) ? z + s[c] : z; }
    float Z = 0.0f, sg = 0.0f;
#pragma unroll
    for (int c = 0; c < CT; c++) {
        s[c] = s[c] / z;
        if (c < C) { Z += s[c] + kMinProb; sg += s[c] * gg[c]; }
    }
#pragma unroll
    for (int c = 0; c < CT; c++)
        if (c < C) dx[base + (size_t)c * HW] = s[c] * (gg[c] - sg) / Z;
}
int launch_softmax_fwd(int B, int C, int HW, const float *x, float *p, hipStream_t stream, float floor_at, float *q0) {
    if (C < 1 || C > kMaxLabels) return set_error(DSRG_ERR_UNSUPPORTED, "1 <= C <= %d required", kMaxLabels);
    const int threads = kSmParts * kSmPix, blocks = (B * HW + kSmPix - 1) / kSmPix;
    if (C <= 21) hipLaunchKernelGGL(softmax_fwd_kernel<21>, dim3(blocks), dim3(threads), 0, stream, B, C, HW, x, p, floor_at, q0);
    else hipLaunchKernelGGL(softmax_fwd_kernel<kMaxLabels>, dim3(blocks), dim3(threads), 0, stream, B, C, HW, x, p, floor_at, q0);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}
int launch_softmax_bwd(int B, int C, int HW, const float *x, const float *g, float *dx, hipStream_t stream) {
    if (C < 1 || C > kMaxLabels) return set_error(DSRG_ERR_UNSUPPORTED, "1 <= C <= %d required", kMaxLabels);
    const int threads = 256, blocks = (B * HW + threads - 1) / threads;
    if (C <= 21) hipLaunchKernelGGL(softmax_bwd_kernel<21>, dim3(blocks), dim3(threads), 0, stream, B, C, HW, x, g, dx);
    else hipLaunchKernelGGL(softmax_bwd_kernel<kMaxLabels>, dim3(blocks), dim3(threads), 0, stream, B, C, HW, x, g, dx);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

// ---- CRFLayer.backward (pylayers.py:90-92) -------------------------------------------------
__global__ void crf_bwd_kernel(size_t n, const double *__restrict__ refined, const float *__restrict__ td,
                               float *__restrict__ bd) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) bd[i] = (float)((1.0 - refined[i]) * (double)td[i]);
}
int launch_crf_bwd(size_t n, const double *refined, const float *td, float *bd, hipStream_t stream) {
    const int threads = 256;
    size_t blocks = (n + threads - 1) / threads;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(crf_bwd_kernel, dim3((unsigned)blocks), dim3(threads), 0, stream, n, refined, td, bd);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

// ---- workgroup reduction helper (double) ---------------------------------------------------
template <int NV>
__device__ __forceinline__ void block_reduce_sum(double (&v)[NV], double *scratch /* [NV*16] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int q = 0; q < NV; q++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[q] += __shfl_down(v[q], off, 64);
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NV; q++) scratch[q * 16 + wave] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; q++) {
        double s = 0.0;
        for (int w = 0; w < nw; w++) s += scratch[q * 16 + w];    // fixed order: deterministic
        v[q] = s;
    }
}

// ---- BalancedSeedLossLayer (pylayers.py:126-152) -------------------------------------------------
// per-image statistics: {count_bg, count_fg, sum S0 log p0, sum S_fg log p_fg}
__device__ __forceinline__ void seed_stats(int C, int HW, const float *__restrict__ pb,
                                           const float *__restrict__ Sb, double (&st)[4], double *scratch) {
    st[0] = st[1] = st[2] = st[3] = 0.0;
    const int n = C * HW;
    // four elements of a thread per round, their loads issued together (one workgroup walks a whole image — or, in the forward
    // kernel, the batch: with one load in flight per thread the stand-alone forward took 0.3 ms for 16 images); a thread adds its
    // elements in index order, as before
    constexpr int U = 4;
    for (int i0 = threadIdx.x; i0 < n; i0 += U * blockDim.x) {
        float sv[U], pv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = min(i0 + u * (int)blockDim.x, n - 1);
            sv[u] = Sb[i];
            pv[u] = pb[i];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = i0 + u * (int)blockDim.x;
            const float s = i < n ? sv[u] : 0.0f;
            if (s != 0.0f) {
                const double t = (double)s * (double)logf(pv[u]);
                if (i < HW) { st[0] += s; st[2] += t; } else { st[1] += s; st[3] += t; }
            }
        }
    }
    block_reduce_sum<4>(st, scratch);
}
// forward: one workgroup walks the images in order (deterministic sum over the batch)
__global__ __launch_bounds__(1024) void seed_loss_fwd_kernel(int B, int C, int HW, const float *__restrict__ p,
                                                             const float *__restrict__ S,
                                                             float *__restrict__ loss) {
    __shared__ double scratch[4 * 16];
    double acc = 0.0;
    for (int b = 0; b < B; b++) {
        double st[4];
        seed_stats(C, HW, p + (size_t)b * C * HW, S + (size_t)b * C * HW, st, scratch);
        const double dbg = st[0] > 1e-4 ? st[0] : 1e-4, dfg = st[1] > 1e-4 ? st[1] : 1e-4;
        acc += -(st[2] / dbg) / B - (st[3] / dfg) / B;
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)acc;
}
// backward: grad = -S / (p * max(count,1e-4) * B); one workgroup per image
__global__ __launch_bounds__(1024) void seed_loss_bwd_kernel(int B, int C, int HW, const float *__restrict__ p,
                                                             const float *__restrict__ S,
                                                             float *__restrict__ grad) {
    __shared__ double scratch[4 * 16];
    const int b = blockIdx.x;
    const float *pb = p + (size_t)b * C * HW, *Sb = S + (size_t)b * C * HW;
    float *gb = grad + (size_t)b * C * HW;
    double st[4];
    seed_stats(C, HW, pb, Sb, st, scratch);
    const float dbg = (float)(st[0] > 1e-4 ? st[0] : 1e-4), dfg = (float)(st[1] > 1e-4 ? st[1] : 1e-4);
    const int n = C * HW;
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        gb[i] = -Sb[i] / (pb[i] * (i < HW ? dbg : dfg) * (float)B);
}
int launch_seed_loss(int B, int C, int HW, const float *p, const float *S, float *loss, float *grad,
                     hipStream_t stream) {
    if (loss) {
        hipLaunchKernelGGL(seed_loss_fwd_kernel, dim3(1), dim3(1024), 0, stream, B, C, HW, p, S, loss);
        DSRG_LAUNCH_CHECK();
    }
    if (grad) {
        hipLaunchKernelGGL(seed_loss_bwd_kernel, dim3(B), dim3(1024), 0, stream, B, C, HW, p, S, grad);
        DSRG_LAUNCH_CHECK();
    }
    return DSRG_OK;
}

// ---- ConstrainLossLayer (pylayers.py:160-180) ------------------------------------------------------
__device__ __forceinline__ float constrain_term(float p, float lq, float &dp, float &dlq) {
    const float q = expf(lq);
    const float r = q / p;
    const bool in = (r >= 0.05f) && (r <= 20.0f);          // T.clip's gradient is 1 on the closed interval
    const float rc = fminf(fmaxf(r, 0.05f), 20.0f);
    const float l = logf(rc);
    dp = in ? -(q / p) : 0.0f;
    dlq = q * (l + (in ? 1.0f : 0.0f));
    return q * l;
}
__global__ __launch_bounds__(1024) void constrain_fwd_kernel(size_t n, double inv, const float *__restrict__ p,
                                                             const float *__restrict__ lq,
                                                             float *__restrict__ loss) {
    __shared__ double scratch[16];
    double acc[1] = {0.0};
    constexpr int U = 4;                                   // as seed_stats: four loads of a thread in flight together
    for (size_t i0 = threadIdx.x; i0 < n; i0 += U * (size_t)blockDim.x) {
        float pv[U], lv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = min(i0 + (size_t)u * blockDim.x, n - 1);
            pv[u] = p[i];
            lv[u] = lq[i];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            float dp, dlq;
            if (i0 + (size_t)u * blockDim.x < n) acc[0] += (double)constrain_term(pv[u], lv[u], dp, dlq);
        }
    }
    block_reduce_sum<1>(acc, scratch);
    if (threadIdx.x == 0) *loss = (float)(acc[0] * inv);
}
__global__ void constrain_bwd_kernel(size_t n, float inv, const float *__restrict__ p,
                                     const float *__restrict__ lq, float *__restrict__ gp,
                                     float *__restrict__ glq) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float dp, dlq;
        constrain_term(p[i], lq[i], dp, dlq);
        if (gp) gp[i] = dp * inv;
        if (glq) glq[i] = dlq * inv;
    }
}
int launch_constrain_loss(int B, int C, int HW, const float *p, const float *lq, float *loss, float *gp,
                          float *glq, hipStream_t stream) {
    const size_t n = (size_t)B * C * HW;
    const double inv = 1.0 / ((double)B * (double)HW);
    if (loss) {
        hipLaunchKernelGGL(constrain_fwd_kernel, dim3(1), dim3(1024), 0, stream, n, inv, p, lq, loss);
        DSRG_LAUNCH_CHECK();
    }
    if (gp || glq) {
        size_t blocks = (n + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(constrain_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, n, (float)inv, p, lq,
                           gp, glq);
        DSRG_LAUNCH_CHECK();
    }
    return DSRG_OK;
}

// ---- fused loss + backward of the five layers (train-s.prototxt:746-810) ------------------
// stage 1: per-image statistics {count_bg, count_fg, seed_bg, seed_fg, constrain}, computed by
// kStatSplit workgroups per image (partials combined in a fixed order by stage 2: deterministic)
constexpr int kStatSplit = 8;
__global__ __launch_bounds__(1024) void sup_stats_kernel(int C, int HW, const float *__restrict__ probs,
                                                         const float *__restrict__ seeds,
                                                         const float *__restrict__ logq,
                                                         double *__restrict__ stats /* [B][kStatSplit][5] */) {
    __shared__ double scratch[5 * 16];
    const int b = blockIdx.x / kStatSplit, part = blockIdx.x % kStatSplit;
    const float *pb = probs + (size_t)b * C * HW, *Sb = seeds + (size_t)b * C * HW, *lb = logq + (size_t)b * C * HW;
    double st[5] = {0, 0, 0, 0, 0};
    const int n = C * HW;
    constexpr int U = 5;                        // elements per thread per batch of loads: 8 x 1024 x 5 >= 21 x 41 x 41, ONE trip per image part (with 4, 7 % of the threads went round twice)
    for (int i0 = part * 1024 * U + threadIdx.x; i0 < n; i0 += kStatSplit * 1024 * U) {
        float s[U], p[U], q[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = min(i0 + u * 1024, n - 1);
            s[u] = Sb[i]; p[u] = pb[i]; q[u] = lb[i];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = i0 + u * 1024;
            if (i < n) {
                if (s[u] != 0.0f) {
                    const double t = (double)s[u] * (double)logf(p[u]);
                    if (i < HW) { st[0] += s[u]; st[2] += t; } else { st[1] += s[u]; st[3] += t; }
                }
                float dp, dlq;
                st[4] += (double)constrain_term(p[u], q[u], dp, dlq);
            }
        }
    }
    block_reduce_sum<5>(st, scratch);
    if (threadIdx.x == 0)
        for (int k = 0; k < 5; k++) stats[((size_t)b * kStatSplit + part) * 5 + k] = st[k];
}
// stage 2: per pixel: total gradient wrt the (clipped) softmax blob, then SoftmaxLayer.backward.  kGradParts adjacent lanes
// share a pixel and split its labels (c = part, part + 4, ...): a thread per pixel walked 21 fp64-rounded exps, logs and
// 105 loads alone, and 26 896 such threads fill a tenth of the chip (20.5 us).  Every sum over the labels is still formed in
// label order (each lane of the pixel repeats it from shuffled terms), so the result is the one-thread result bit for bit.
// Block 0 also finalises the two loss scalars: per-image terms in parallel, their sum in image order.
constexpr int kGradParts = 4, kGradPix = 64;
template <int CT>
__global__ __launch_bounds__(kGradParts * kGradPix) void sup_grad_kernel(int B, int C, int HW, const float *__restrict__ logits,
                                                       const float *__restrict__ probs,
                                                       const float *__restrict__ seeds,
                                                       const float *__restrict__ logq,
                                                       const double *__restrict__ refined,
                                                       const double *__restrict__ stats,
                                                       float *__restrict__ grad_logits,
                                                       float *__restrict__ losses) {
    constexpr int P = kGradParts, LPP = (CT + P - 1) / P;       // labels per part
    const int lane = threadIdx.x & 63, part = threadIdx.x & (P - 1), lane0 = lane & ~(P - 1);
    const int idx = blockIdx.x * kGradPix + (int)(threadIdx.x / P);
    const bool live = idx < B * HW;
    const int b = live ? idx / HW : 0, i = live ? idx - b * HW : 0;
    const size_t base = (size_t)b * C * HW + i;
    // loads first: everything this lane will need, unconditionally (clamped label index)
    float s[LPP], pv[LPP], lq[LPP], sd[LPP];
    double rf[LPP];
#pragma unroll
    for (int t = 0; t < LPP; t++) {
        const size_t o = base + (size_t)min(part + t * P, C - 1) * HW;
        s[t] = logits[o]; pv[t] = probs[o]; lq[t] = logq[o]; sd[t] = seeds[o]; rf[t] = refined[o];
    }
    double cnt_bg = 0.0, cnt_fg = 0.0;
    for (int sp = 0; sp < kStatSplit; sp++) {
        cnt_bg += stats[((size_t)b * kStatSplit + sp) * 5 + 0];
        cnt_fg += stats[((size_t)b * kStatSplit + sp) * 5 + 1];
    }
    if (blockIdx.x == 0) {
        __shared__ double term_s[kGradParts * kGradPix], term_c[kGradParts * kGradPix];
        double ls = 0.0, lc = 0.0;
        for (int b0 = 0; b0 < B; b0 += kGradParts * kGradPix) {
            const int bb = b0 + (int)threadIdx.x;
            if (bb < B) {
                double st[5] = {0, 0, 0, 0, 0};
                for (int sp = 0; sp < kStatSplit; sp++)
                    for (int k = 0; k < 5; k++) st[k] += stats[((size_t)bb * kStatSplit + sp) * 5 + k];
                const double dbg = st[0] > 1e-4 ? st[0] : 1e-4, dfg = st[1] > 1e-4 ? st[1] : 1e-4;
                term_s[threadIdx.x] = -(st[2] / dbg) / B - (st[3] / dfg) / B;
                term_c[threadIdx.x] = st[4];
            }
            __syncthreads();
            if (threadIdx.x == 0)
                for (int k = 0; k < min(kGradParts * kGradPix, B - b0); k++) { ls += term_s[k]; lc += term_c[k]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            losses[0] = (float)ls;
            losses[1] = (float)(lc / ((double)B * (double)HW));
        }
    }
    const float dbg = (float)(cnt_bg > 1e-4 ? cnt_bg : 1e-4), dfg = (float)(cnt_fg > 1e-4 ? cnt_fg : 1e-4);
    const float inv = (float)(1.0 / ((double)B * (double)HW));
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < LPP; t++) { s[t] = (part + t * P < C) ? s[t] : -INFINITY; mx = fmaxf(mx, s[t]); }
#pragma unroll
    for (int m = 1; m < P; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    float g[LPP];
#pragma unroll
    for (int t = 0; t < LPP; t++) {
        const int c = part + t * P;
        s[t] = exp_cr(s[t] - mx);
        // BalancedSeedLoss.backward + ConstrainLoss.backward[0] + CRFLayer.backward(ConstrainLoss.backward[1])
        float dp, dlq;
        constrain_term(pv[t], lq[t], dp, dlq);
        const float gseed = -sd[t] / (pv[t] * (c == 0 ? dbg : dfg) * (float)B);
        const float gcrf = (float)((1.0 - rf[t]) * (double)(dlq * inv));
        g[t] = gseed + dp * inv + gcrf;
    }
    float z = 0.0f;                                  // label-order sums: term c comes from lane part c % 4, slot c / 4
#pragma unroll
    for (int t = 0; t < LPP; t++)
#pragma unroll
        for (int q = 0; q < P; q++) {
            const float e = __shfl(s[t], lane0 + q, 64);
            if (t * P + q < C) z = z + e;
        }
    float Z = 0.0f, sg = 0.0f;
#pragma unroll
    for (int t = 0; t < LPP; t++) s[t] = s[t] / z;
#pragma unroll
    for (int t = 0; t < LPP; t++)
#pragma unroll
        for (int q = 0; q < P; q++) {
            const float sv = __shfl(s[t], lane0 + q, 64), pr = __shfl(s[t] * g[t], lane0 + q, 64);
            if (t * P + q < C) { Z += sv + kMinProb; sg += pr; }
        }
    if (!live) return;
#pragma unroll
    for (int t = 0; t < LPP; t++) {
        const int c = part + t * P;
        if (c < C) grad_logits[base + (size_t)c * HW] = s[t] * (g[t] - sg) / Z;
    }
}
int launch_sup_loss_backward(int B, int C, int HW, const float *logits, const float *probs, const float *seeds,
                             const float *logq, const double *refined, double *stats, float *grad_logits,
                             float *losses, hipStream_t stream) {
    if (C < 1 || C > kMaxLabels) return set_error(DSRG_ERR_UNSUPPORTED, "1 <= C <= %d required", kMaxLabels);
    hipLaunchKernelGGL(sup_stats_kernel, dim3(B * kStatSplit), dim3(1024), 0, stream, C, HW, probs, seeds, logq, stats);
    DSRG_LAUNCH_CHECK();
    const int threads = kGradParts * kGradPix, blocks = (B * HW + kGradPix - 1) / kGradPix;
    if (C <= 21)
        hipLaunchKernelGGL(sup_grad_kernel<21>, dim3(blocks), dim3(threads), 0, stream, B, C, HW, logits, probs, seeds,
                           logq, refined, stats, grad_logits, losses);
    else
        hipLaunchKernelGGL(sup_grad_kernel<kMaxLabels>, dim3(blocks), dim3(threads), 0, stream, B, C, HW, logits,
                           probs, seeds, logq, refined, stats, grad_logits, losses);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

// ---- layout helpers for the single-image (host pointer, label-fastest) API ---------------------------
// in [N][M] label-fastest -> out [M][N] planes, optionally negated
__global__ void lf_to_planes_kernel(int N, int M, const float *__restrict__ in, float *__restrict__ out, int negate) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * M) return;
    const int c = idx / N, i = idx - c * N;
    const float v = in[(size_t)i * M + c];
    out[idx] = negate ? -v : v;
}
__global__ void planes_to_lf_kernel(int N, int M, const float *__restrict__ in, float *__restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * M) return;
    const int i = idx / M, c = idx - i * M;
    out[idx] = in[(size_t)c * N + i];
}
__global__ void argmax_planes_kernel(int N, int M, const float *__restrict__ q, int32_t *__restrict__ lab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int m = 0;
    float best = q[i];
    for (int c = 1; c < M; c++) {
        const float v = q[(size_t)c * N + i];
        if (v > best) { best = v; m = c; }      // first maximum wins (Eigen maxCoeff, densecrf.cpp:136-140)
    }
    lab[i] = m;
}
int launch_lf_to_planes(int N, int M, const float *in, float *out, int negate, hipStream_t stream) {
    hipLaunchKernelGGL(lf_to_planes_kernel, dim3((N * M + 255) / 256), dim3(256), 0, stream, N, M, in, out, negate);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}
int launch_planes_to_lf(int N, int M, const float *in, float *out, hipStream_t stream) {
    hipLaunchKernelGGL(planes_to_lf_kernel, dim3((N * M + 255) / 256), dim3(256), 0, stream, N, M, in, out);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}
int launch_argmax_planes(int N, int M, const float *q, int32_t *lab, hipStream_t stream) {
    hipLaunchKernelGGL(argmax_planes_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, N, M, q, lab);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

}  // namespace dsrg

// ---- backbone plumbing: NHWC im2col for 3x3 stride-1 "same" (dilated) convolutions, 2-byte elements ----------
// out[(b,y,x)][tap][c] = in[b][y + (ty-1)*dil][x + (tx-1)*dil][c] (zero outside); one thread moves 16 bytes.
// Feeds one hipBLASLt GEMM per layer (dsrg_amd/backbone.py); torch.cat needs 0.28 ms per 41x41x512 layer
// for the same copy, this kernel is bandwidth-bound.
namespace dsrg {
__global__ void im2col3x3_nhwc16_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, int B, int H, int W,
                                        int C8, int dil) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * H * W * 9 * C8;
    if (idx >= total) return;
    const int c = (int)(idx % C8);
    size_t r = idx / C8;
    const int tap = (int)(r % 9);
    r /= 9;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    const int b = (int)(r / H);
    const int yy = y + (tap / 3 - 1) * dil, xx = x + (tap % 3 - 1) * dil;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = in[(((size_t)b * H + yy) * W + xx) * C8 + c];
    // nontemporal: the patch matrix (248 MB for a 41x41x512 layer at batch 16) is far larger than the L2s and is read once by
    // the GEMM that follows; keeping it out of them left the train step 1.5 % faster (1 200 -> 1 219 img/s)
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<u4 *>(out) + idx);
}
// adjoint of the im2col above: out[b,y,x,:] = sum over the 9 taps of cols[(b, y - dy*dil, x - dx*dil), tap, :] for the source
// pixels that exist (dy, dx in {-1,0,1}); cols is (B*H*W, 9*C) 2-byte bf16, sums in f32.  Used for the data gradient of a
// 3x3 convolution with more output than input channels: g @ W^T first (no im2col of the wide g), then this gather.
__global__ void col2im3x3_nhwc16_kernel(const uint4 *__restrict__ cols, uint4 *__restrict__ out, int B, int H, int W,
                                        int C8, int dil) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * H * W * C8;
    if (idx >= total) return;
    const int c = (int)(idx % C8);
    size_t r = idx / C8;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    const int b = (int)(r / H);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int yy = y - (tap / 3 - 1) * dil, xx = x - (tap % 3 - 1) * dil;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        const uint4 v = cols[((((size_t)b * H + yy) * W + xx) * 9 + tap) * C8 + c];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[2 * k] += __uint_as_float(w[k] << 16);
            acc[2 * k + 1] += __uint_as_float(w[k] & 0xffff0000u);
        }
    }
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t lo = __float_as_uint(acc[2 * k]), hi = __float_as_uint(acc[2 * k + 1]);
        lo = (lo + 0x7fffu + ((lo >> 16) & 1u)) >> 16;                    // round to nearest even (finite sums)
        hi = (hi + 0x7fffu + ((hi >> 16) & 1u)) >> 16;
        o[k] = lo | (hi << 16);
    }
    out[idx] = make_uint4(o[0], o[1], o[2], o[3]);
}
int launch_col2im3x3(const void *cols, void *out, int B, int H, int W, int C, int dil, hipStream_t stream) {
    if (C % 8 != 0) return set_error(DSRG_ERR_UNSUPPORTED, "col2im: channels must be a multiple of 8");
    const size_t total = (size_t)B * H * W * (C / 8);
    hipLaunchKernelGGL(col2im3x3_nhwc16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       (const uint4 *)cols, (uint4 *)out, B, H, W, C / 8, dil);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

int launch_im2col3x3(const void *in, void *out, int B, int H, int W, int C, int dil, hipStream_t stream) {
    if (C % 8 != 0) return set_error(DSRG_ERR_UNSUPPORTED, "im2col: channels must be a multiple of 8");
    const size_t total = (size_t)B * H * W * 9 * (C / 8);
    hipLaunchKernelGGL(im2col3x3_nhwc16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                       (const uint4 *)in, (uint4 *)out, B, H, W, C / 8, dil);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}
}  // namespace dsrg
