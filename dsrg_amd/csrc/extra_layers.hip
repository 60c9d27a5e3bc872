// The pylayers classes no seed_mc prototxt references, and the evaluation histogram:
//   SeedLossLayer        pylayers/pylayers/pylayers.py:94-118
//   ExpandLossLayer      pylayers/pylayers/pylayers.py:183-233   (SEC's global weighted rank pooling: a sort per label plane)
//   ConfusionMatrix.add / generateM   training/tools/evaluate.py:25-30,61-68
#include "common.h"

namespace dsrg {

// dynamic LDS next to static arrays: above 48 KB ask for the size explicitly (the default cap is 64 KB for both together)
static int reserve_lds(const void *fn, size_t bytes, LdsGrant &grant) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    std::atomic<size_t> &granted = grant.bytes[dev];
    size_t have = granted.load(std::memory_order_acquire);
    if (bytes <= have) return DSRG_OK;
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess)
            return set_error(DSRG_ERR_HIP, "cannot reserve %zu B of dynamic LDS: %s", bytes, hipGetErrorString(e));
    }
    while (have < bytes && !granted.compare_exchange_weak(have, bytes, std::memory_order_release, std::memory_order_acquire)) {}
    return DSRG_OK;
}

// fixed-order workgroup sum of NV doubles (wave shuffles, then the wave partials in wave order)
template <int NV>
__device__ __forceinline__ void wg_sum(double (&v)[NV], double *scratch /* [NV*16] */) {
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
        for (int w = 0; w < nw; w++) s += scratch[q * 16 + w];
        v[q] = s;
    }
    __syncthreads();
}

// ---- SeedLossLayer ----------------------------------------------------------------------------------------------
// per image: {count, sum S log p}; no floor on the count (an empty seed map divides by zero exactly as Theano's graph)
__device__ __forceinline__ void plain_seed_stats(int n, const float *__restrict__ pb, const float *__restrict__ Sb,
                                                 double (&st)[2], double *scratch) {
    st[0] = st[1] = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float s = Sb[i];
        if (s != 0.0f) { st[0] += s; st[1] += (double)s * (double)logf(pb[i]); }
    }
    wg_sum<2>(st, scratch);
}
__global__ __launch_bounds__(1024) void seed_plain_fwd_kernel(int B, int n, const float *__restrict__ p,
                                                              const float *__restrict__ S, float *__restrict__ loss) {
    __shared__ double scratch[2 * 16];
    double acc = 0.0;
    for (int b = 0; b < B; b++) {                                    // images in order: deterministic batch sum
        double st[2];
        plain_seed_stats(n, p + (size_t)b * n, S + (size_t)b * n, st, scratch);
        acc += -(st[1] / st[0]) / B;
    }
    if (threadIdx.x == 0) *loss = (float)acc;
}
__global__ __launch_bounds__(1024) void seed_plain_bwd_kernel(int B, int n, const float *__restrict__ p,
                                                              const float *__restrict__ S, float *__restrict__ grad) {
    __shared__ double scratch[2 * 16];
    const float *pb = p + (size_t)blockIdx.x * n, *Sb = S + (size_t)blockIdx.x * n;
    float *gb = grad + (size_t)blockIdx.x * n;
    double st[2];
    plain_seed_stats(n, pb, Sb, st, scratch);
    const double cnt = st[0];
    for (int i = threadIdx.x; i < n; i += blockDim.x) gb[i] = (float)(-(double)Sb[i] / ((double)pb[i] * cnt * B));
}
int launch_seed_loss_plain(int B, int C, int HW, const float *p, const float *S, float *loss, float *grad, hipStream_t stream) {
    if (loss) {
        hipLaunchKernelGGL(seed_plain_fwd_kernel, dim3(1), dim3(1024), 0, stream, B, C * HW, p, S, loss);
        DSRG_LAUNCH_CHECK();
    }
    if (grad) {
        hipLaunchKernelGGL(seed_plain_bwd_kernel, dim3(B), dim3(1024), 0, stream, B, C * HW, p, S, grad);
        DSRG_LAUNCH_CHECK();
    }
    return DSRG_OK;
}

// ---- ExpandLossLayer ---------------------------------------------------------------------------------------------
// One workgroup per (image, label plane).  Pooled planes (background, present foreground): the plane's values are
// packed with their pixel index into 64-bit keys, bitonic-sorted in LDS (ties: lower pixel index first, like a stable
// sort), weighted by q^(HW-1-rank) in fp64; absent foreground planes only need their maximum.  Each workgroup writes
// its plane's gradient and its term of the loss; a second launch adds the terms in (image, label) order.
constexpr int kExpandMaxHW = 8192;

__device__ __forceinline__ uint32_t float_order_bits(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(1024) void expand_plane_kernel(int B, int C, int HW, int NP, const float *__restrict__ p,
                                                            const float *__restrict__ stat, double q_fg, double q_bg,
                                                            double *__restrict__ terms, float *__restrict__ grad) {
    extern __shared__ unsigned long long keys[];                       // [NP]
    __shared__ double scratch[2 * 16];
    __shared__ float s_max[16];
    const int b = blockIdx.x / C, c = blockIdx.x % C;
    const float *pl = p + (size_t)blockIdx.x * HW;
    float *gl = grad ? grad + (size_t)blockIdx.x * HW : nullptr;
    // image-level label counts over the foreground classes (stat[:, 0] is not read: pylayers.py:193)
    int n_pres = 0;
    for (int k = 1; k < C; k++) n_pres += stat[(size_t)b * C + k] > 0.5f ? 1 : 0;
    const int n_abs = (C - 1) - n_pres;
    const bool present = c > 0 && stat[(size_t)b * C + c] > 0.5f;

    if (c == 0 || present) {
        for (int i = threadIdx.x; i < NP; i += blockDim.x)
            keys[i] = i < HW ? ((unsigned long long)float_order_bits(pl[i]) << 32) | (unsigned)i : ~0ull;
        __syncthreads();
        for (int k = 2; k <= NP; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = threadIdx.x; i < NP; i += blockDim.x) {
                    const int l = i ^ j;
                    if (l > i) {
                        const unsigned long long a = keys[i], d = keys[l];
                        const bool up = (i & k) == 0;
                        if ((a > d) == up) { keys[i] = d; keys[l] = a; }
                    }
                }
                __syncthreads();
            }
        }
        const double q = c == 0 ? q_bg : q_fg;
        double st[2] = {0.0, 0.0};                                     // {sum of weights, sum of value * weight}
        for (int k = threadIdx.x; k < HW; k += blockDim.x) {
            const double w = pow(q, (double)(HW - 1 - k));
            st[0] += w;
            st[1] += (double)pl[(unsigned)keys[k]] * w;
        }
        wg_sum<2>(st, scratch);
        const double z = st[0], pooled = st[1] / z;
        const double coef = c == 0 ? 1.0 / B : 1.0 / ((double)n_pres * B);
        if (threadIdx.x == 0 && terms) terms[blockIdx.x] = -log(pooled) * coef;
        if (gl) {
            for (int k = threadIdx.x; k < HW; k += blockDim.x)
                gl[(unsigned)keys[k]] = (float)(-coef / pooled * (pow(q, (double)(HW - 1 - k)) / z));
        }
    } else {
        float mx = -__builtin_inff();
        for (int i = threadIdx.x; i < HW; i += blockDim.x) mx = fmaxf(mx, pl[i]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_down(mx, off, 64));
        if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = mx;
        __syncthreads();
        mx = s_max[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) mx = fmaxf(mx, s_max[w]);
        const double coef = 1.0 / ((double)n_abs * B);
        if (threadIdx.x == 0 && terms) terms[blockIdx.x] = -log(1.0 - (double)mx) * coef;
        if (gl) {
            const float g = (float)(coef / (1.0 - (double)mx));
            for (int i = threadIdx.x; i < HW; i += blockDim.x) gl[i] = pl[i] == mx ? g : 0.0f;   // every tied pixel (eq mask)
        }
    }
}
__global__ void expand_total_kernel(int n, const double *__restrict__ terms, float *__restrict__ loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n; i++) s += terms[i];
        *loss = (float)s;
    }
}
int launch_expand_loss(int B, int C, int HW, const float *p, const float *stat, double q_fg, double q_bg, float *loss,
                       float *grad, double *terms, hipStream_t stream) {
    if (HW > kExpandMaxHW) return set_error(DSRG_ERR_UNSUPPORTED, "expand loss: label planes above 8192 pixels");
    if (C < 2) return set_error(DSRG_ERR_INVALID, "expand loss: needs a background and at least one foreground label");
    if (loss && !terms) return set_error(DSRG_ERR_INVALID, "expand loss: the loss needs the B*C scratch doubles");
    int NP = 1;
    while (NP < HW) NP <<= 1;
    static LdsGrant granted;
    int rc = reserve_lds((const void *)expand_plane_kernel, (size_t)NP * 8, granted);
    if (rc) return rc;
    hipLaunchKernelGGL(expand_plane_kernel, dim3(B * C), dim3(1024), (size_t)NP * 8, stream, B, C, HW, NP, p, stat, q_fg, q_bg,
                       loss ? terms : nullptr, grad);
    DSRG_LAUNCH_CHECK();
    if (loss) {
        hipLaunchKernelGGL(expand_total_kernel, dim3(1), dim3(64), 0, stream, B * C, terms, loss);
        DSRG_LAUNCH_CHECK();
    }
    return DSRG_OK;
}

// ---- confusion matrix ----------------------------------------------------------------------------------------------
// hist[gt * nclass + pred] += 1 over the pixels whose ground truth passes the rule; LDS histogram per workgroup, 64-bit
// global counters.  hist[nclass * nclass] counts predictions >= nclass (the reference asserts there are none).
__global__ __launch_bounds__(256) void confusion_kernel(size_t n, const unsigned char *__restrict__ gt,
                                                        const unsigned char *__restrict__ pred, int nclass, int rule_lt,
                                                        unsigned long long *__restrict__ hist) {
    extern __shared__ unsigned int h[];                                // [nclass*nclass + 1]
    const int nb = nclass * nclass + 1;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) h[i] = 0u;
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int g = gt[i], q = pred[i];
        const bool keep = rule_lt ? g < nclass : g != 255;
        if (!keep) continue;
        if (q >= nclass || g >= nclass) atomicAdd(&h[nb - 1], 1u);
        else atomicAdd(&h[g * nclass + q], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += blockDim.x)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}
int launch_confusion(size_t n, const unsigned char *gt, const unsigned char *pred, int nclass, int rule_lt,
                     unsigned long long *hist, hipStream_t stream) {
    if (nclass < 1 || nclass > 127) return set_error(DSRG_ERR_UNSUPPORTED, "confusion matrix: 1..127 classes");
    if (n == 0) return DSRG_OK;
    const size_t lds = ((size_t)nclass * nclass + 1) * sizeof(unsigned int);
    static LdsGrant granted;
    int rc = reserve_lds((const void *)confusion_kernel, lds, granted);
    if (rc) return rc;
    size_t blocks = (n + 256 * 16 - 1) / (256 * 16);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, n, gt, pred, nclass, rule_lt, hist);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

}  // namespace dsrg
