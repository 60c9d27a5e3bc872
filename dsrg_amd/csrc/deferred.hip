// Deferred bias-gradient reductions (include/dsrg_hip.h: dsrg_defer_reductions / dsrg_flush_reductions).
//
// Fifteen launches of a train-s step end with a pass of 5-7 us that turns per-tile / per-block partial rows into a bias gradient
// (igemm_colsum_kernel x 10, bias_finalize_kernel x 5: 96 us of an 8.2 ms step) and nothing reads a bias gradient before the
// update.  While deferral is on those passes are recorded (common.h: defer_reduction) and dsrg_flush_reductions runs them all in
// ONE launch — the same device bodies on the same partial rows in the same order: the gradients' bits do not change.  The caller
// keeps every recorded launch's partial rows alive and unshared until the flush (dsrg_amd/ops.py does).
#include "common.h"
#include <mutex>
#include <cstring>

namespace dsrg {
namespace {

constexpr int kMaxDeferred = 48;
struct DeferredReduction {
    const float *part;
    float *out;
    int rows, cols, kind, first_block;
};
struct DeferredArgs {
    DeferredReduction e[kMaxDeferred];
    int n;
};

__global__ __launch_bounds__(1024) void deferred_reductions_kernel(DeferredArgs a) {
    __shared__ float red_c[16][64];
    __shared__ float red_b[8][33];
    const int b = (int)blockIdx.x;
    int k = 0;
    while (k + 1 < a.n && b >= a.e[k + 1].first_block) k++;              // (block-uniform)
    const DeferredReduction d = a.e[k];
    if (d.kind == 0) colsum_block(d.part, d.out, d.rows, d.cols, b - d.first_block, red_c);
    else bias_finalize_block(d.part, d.out, d.rows, d.cols, b - d.first_block, red_b);
}

std::mutex g_mutex;
bool g_on = false;
DeferredArgs g_list;
int g_blocks = 0;

}  // namespace

bool defer_reduction(int kind, const float *part, float *out, int rows, int cols) {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (!g_on || g_list.n >= kMaxDeferred || rows < 1 || cols < 1 || (kind == 0 && cols % 64)) return false;
    DeferredReduction &d = g_list.e[g_list.n++];
    d.part = part; d.out = out; d.rows = rows; d.cols = cols; d.kind = kind; d.first_block = g_blocks;
    g_blocks += kind == 0 ? cols / 64 : (cols + 31) / 32;
    return true;
}

int defer_reductions(int on) {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (on && !g_on) { memset(&g_list, 0, sizeof(g_list)); g_blocks = 0; }
    g_on = on != 0;
    return DSRG_OK;
}

int flush_reductions(hipStream_t stream) {
    DeferredArgs a;
    int blocks;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        a = g_list;
        blocks = g_blocks;
        memset(&g_list, 0, sizeof(g_list));
        g_blocks = 0;
    }
    if (a.n == 0) return DSRG_OK;
    hipLaunchKernelGGL(deferred_reductions_kernel, dim3(blocks), dim3(1024), 0, stream, a);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

}  // namespace dsrg

extern "C" int dsrg_defer_reductions(int on) { return dsrg::defer_reductions(on); }
extern "C" int dsrg_flush_reductions(void *stream) { return dsrg::flush_reductions(static_cast<hipStream_t>(stream)); }
