// Caffe's SGD update of a whole list of float32 parameters in one launch per sixteen tensors, with the packed bf16 forms of the
// convolution kernels written in the same pass — backbone plumbing of the PyTorch trainer (solver-s.prototxt:5-14 is Caffe's own
// SGDSolver; no reference counterpart).
//
//   B <- m B + (g + wd W);   W <- W - lr B          (trainer.CaffeSGD's form of  V <- m V + lr (g + wd W);  W <- W - V)
//
// Why it exists: the step ran torch's fused SGD (5 launches, 757 MB) and then, layer by layer inside the next forward, 20 launches of
// pack_conv_weight_kernel that read the 148 MB of wide-layer weights AGAIN to cast and re-lay them out for the implicit-GEMM / direct
// kernels (0.31 ms per step together).  Here a block that updates a 64 x 64 x tap block of a kernel holds its new values in
// registers: it writes them as the forward packing (and, through an LDS transpose, as the data-gradient packing) while they are
// there.  With lr = 0 momentum = 1 and no gradient the same kernel only packs (first step, weights loaded from a file).
//
// Layouts (see conv_igemm.hip, pack_conv_weight_kernel, which this kernel's packing half restates): parameter [o][tap][c]
// (channels_last (cout, cin, k, k)); igemm forward fwd[o][c / 64][tap][64], igemm data gradient dg[c][o / 64][T - tap][64]; plain
// (direct kernels): fwd[o][tap][c], dg[c][T - tap][o].
#include "common.h"
#include <cstring>

namespace dsrg {
namespace {
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {      // v_cvt_pk_bf16_f32: round to nearest even
    f32x2_t v = {lo, hi};
    bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
    return *reinterpret_cast<uint32_t *>(&b);
}

constexpr int kSgdTensors = 16;
struct SgdTensor {
    float *p;               // parameter, updated in place
    const float *g;         // gradient, or null: no update (pack only)
    float *buf;             // momentum buffer B, updated in place (null with g)
    uint16_t *fwd, *dg;     // packed bf16 forms or null
    uint32_t n;             // elements
    uint32_t cout, cin, taps;   // of a packed kernel
    uint32_t first_block;   // this tensor's first block in the launch
    float lr, wd;           // the tensor's group: base_lr * lr_mult, weight_decay * decay_mult
    int plain;              // packed layouts of the direct kernels instead of the implicit-GEMM ones; tensor not packed: 1 = take the scalar path
    int g_split;            // the gradient is only 4-byte aligned (a view into a DistributedDataParallel bucket): dword loads
};
struct SgdArgs {
    SgdTensor t[kSgdTensors];
    int n;
    float momentum;
};

__device__ __forceinline__ float4 sgd4(float4 w, float4 g, float4 &b, float m, float lr, float wd) {
    // the operation order of torch's fused kernel: grad + wd * param, then momentum * buf + that, then param - lr * buf
    float4 d = make_float4(g.x + wd * w.x, g.y + wd * w.y, g.z + wd * w.z, g.w + wd * w.w);
    b = make_float4(m * b.x + d.x, m * b.y + d.y, m * b.z + d.z, m * b.w + d.w);
    return make_float4(w.x - lr * b.x, w.y - lr * b.y, w.z - lr * b.z, w.w - lr * b.w);
}

__device__ __forceinline__ float4 load_grad4(const float *g, int split) {
    if (split) return make_float4(g[0], g[1], g[2], g[3]);    // a lane still reads its own 16 contiguous bytes: the lines are used whole
    return *reinterpret_cast<const float4 *>(g);
}

__global__ __launch_bounds__(256) void sgd_pack_kernel(SgdArgs a) {
    __shared__ uint16_t tile[64][64 + 4];
    // the tensor of this block: the last one whose first block is not beyond it (at most sixteen: a scalar scan)
    int ti = 0;
    for (int k = 1; k < a.n; k++)
        if (a.t[k].first_block <= blockIdx.x) ti = k;
    const SgdTensor T = a.t[ti];
    const uint32_t blk = blockIdx.x - T.first_block;
    const int t = threadIdx.x;
    const float m = a.momentum, lr = T.lr, wd = T.wd;
    if (!T.fwd && !T.dg) {
        // ---- a plain tensor (bias, classifier, first layer): 4096 elements per block, float4 per thread and pass
        if (!T.g) return;
        const size_t base = (size_t)blk * 4096;
        if (T.plain) {                                       // a slice of a larger buffer that is not 16-byte aligned: one float at a time
            for (size_t q = base + t; q < T.n && q < base + 4096; q += 256) {
                const float d = T.g[q] + wd * T.p[q], b = m * T.buf[q] + d;
                T.buf[q] = b;
                T.p[q] = T.p[q] - lr * b;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const size_t e = base + (size_t)(i * 256 + t) * 4;
            if (e + 3 < T.n) {
                const float4 w = *reinterpret_cast<const float4 *>(T.p + e), g = load_grad4(T.g + e, T.g_split);
                float4 b = *reinterpret_cast<const float4 *>(T.buf + e);
                const float4 nw = sgd4(w, g, b, m, lr, wd);
                *reinterpret_cast<float4 *>(T.buf + e) = b;
                *reinterpret_cast<float4 *>(T.p + e) = nw;
            } else {
                for (size_t q = e; q < T.n && q < e + 4; q++) {
                    const float d = T.g[q] + wd * T.p[q], b = m * T.buf[q] + d;
                    T.buf[q] = b;
                    T.p[q] = T.p[q] - lr * b;
                }
            }
        }
        return;
    }
    // ---- a packed kernel: one block per (64 outputs, 64 inputs, tap), as pack_conv_weight_kernel
    const uint32_t cbs = T.cin >> 6, obs = T.cout >> 6;
    const uint32_t tap = blk % T.taps, cb = (blk / T.taps) % cbs, ob = blk / (T.taps * cbs);
    if (ob >= obs) return;
    const int r = t >> 4, q = t & 15;                        // 16 rows x 16 float4 per pass
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t o = r + 16 * i;
        const size_t src = ((size_t)(ob * 64 + o) * T.taps + tap) * T.cin + cb * 64 + q * 4;
        float4 w = *reinterpret_cast<const float4 *>(T.p + src);
        if (T.g) {
            const float4 g = load_grad4(T.g + src, T.g_split);
            float4 b = *reinterpret_cast<const float4 *>(T.buf + src);
            w = sgd4(w, g, b, m, lr, wd);
            *reinterpret_cast<float4 *>(T.buf + src) = b;
            *reinterpret_cast<float4 *>(T.p + src) = w;
        }
        const uint2 pk = make_uint2(pack2_bf16(w.x, w.y), pack2_bf16(w.z, w.w));
        if (T.fwd)
            *reinterpret_cast<uint2 *>(T.fwd + (T.plain ? src : (((size_t)(ob * 64 + o) * cbs + cb) * T.taps + tap) * 64 + q * 4)) = pk;
        *reinterpret_cast<uint2 *>(&tile[o][q * 4]) = pk;
    }
    if (!T.dg) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t c = r + 16 * i;                       // row of the transposed tile; q*4 .. q*4+3 = its outputs
        const uint32_t lo = (uint32_t)tile[q * 4 + 0][c] | ((uint32_t)tile[q * 4 + 1][c] << 16);
        const uint32_t hi = (uint32_t)tile[q * 4 + 2][c] | ((uint32_t)tile[q * 4 + 3][c] << 16);
        const size_t dst = T.plain ? ((size_t)(cb * 64 + c) * T.taps + (T.taps - 1 - tap)) * T.cout + ob * 64 + q * 4
                                   : (((size_t)(cb * 64 + c) * obs + ob) * T.taps + (T.taps - 1 - tap)) * 64 + q * 4;
        *reinterpret_cast<uint2 *>(T.dg + dst) = make_uint2(lo, hi);
    }
}
}  // namespace

// n tensors (any number: sixteen per launch).  p / g / buf: n pointers each (g[i] = buf[i] = null: pack only); fwd / dg: the packed
// forms or null; shape[i] = {cout, cin, taps, plain} for packed tensors (64 | cout, 64 | cin), ignored otherwise; numel[i]
// elements; lr[i], wd[i] the tensor's rates.
int launch_sgd_pack(int n, float *const *p, const float *const *g, float *const *buf, void *const *fwd, void *const *dg,
                    const int *shape, const long long *numel, const float *lr, const float *wd, float momentum, hipStream_t stream) {
    if (n < 0) return set_error(DSRG_ERR_INVALID, "sgd_pack: bad count");
    for (int i0 = 0; i0 < n; i0 += kSgdTensors) {
        SgdArgs a;
        memset(&a, 0, sizeof(a));
        a.momentum = momentum;
        uint32_t blocks = 0;
        for (int i = i0; i < n && i < i0 + kSgdTensors; i++) {
            SgdTensor &T = a.t[a.n++];
            T.p = p[i]; T.g = g ? g[i] : nullptr; T.buf = buf ? buf[i] : nullptr;
            T.fwd = fwd ? static_cast<uint16_t *>(fwd[i]) : nullptr; T.dg = dg ? static_cast<uint16_t *>(dg[i]) : nullptr;
            T.lr = lr ? lr[i] : 0.0f; T.wd = wd ? wd[i] : 0.0f;
            if (!T.p || numel[i] < 1 || numel[i] > 0x7fffffffLL || (T.g && !T.buf))
                return set_error(DSRG_ERR_INVALID, "sgd_pack: tensor %d: null parameter, bad size or a gradient without a momentum buffer", i);
            const uintptr_t low = reinterpret_cast<uintptr_t>(T.p) | reinterpret_cast<uintptr_t>(T.buf);
            if (((low | reinterpret_cast<uintptr_t>(T.g)) & 3) || ((low & 15) && (T.fwd || T.dg)))
                return set_error(DSRG_ERR_INVALID, "sgd_pack: tensor %d is not aligned (4 bytes; parameter and momentum of a packed kernel: 16)", i);
            T.g_split = (reinterpret_cast<uintptr_t>(T.g) & 15) != 0;
            T.n = (uint32_t)numel[i];
            T.first_block = blocks;
            if (T.fwd || T.dg) {
                T.cout = (uint32_t)shape[4 * i]; T.cin = (uint32_t)shape[4 * i + 1]; T.taps = (uint32_t)shape[4 * i + 2]; T.plain = shape[4 * i + 3];
                if (T.cout < 64 || T.cout % 64 || T.cin < 64 || T.cin % 64 || (T.taps != 1 && T.taps != 9) ||
                    (long long)T.cout * T.cin * T.taps != numel[i])
                    return set_error(DSRG_ERR_INVALID, "sgd_pack: tensor %d: a packed kernel needs 64 | cout, 64 | cin, 1 or 9 taps (got %u, %u, %u)",
                                     i, T.cout, T.cin, T.taps);
                blocks += (T.cout / 64) * (T.cin / 64) * T.taps;
            } else {
                T.plain = (low & 15) != 0;
                blocks += (uint32_t)((numel[i] + 4095) / 4096);
            }
        }
        hipLaunchKernelGGL(sgd_pack_kernel, dim3(blocks), dim3(256), 0, stream, a);
        DSRG_LAUNCH_CHECK();
    }
    return DSRG_OK;
}

}  // namespace dsrg
