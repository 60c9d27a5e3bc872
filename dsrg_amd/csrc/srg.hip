// Seeded region growing on gfx950: one workgroup per image, label map and growth
// front resident in LDS with a one-pixel halo.
//
// Replaces generate_seed_step (pylayers/pylayers/pylayers.py:237-275) including
// the per-class connected-component labelling it calls
// (pylayers/pylayers/CC_labeling_8.py:112-197).  The reference labels the
// components of every class mask and keeps the ones that contain a cue; because a
// pixel belongs to exactly one class mask (its label-map value), this equals one
// simultaneous flood fill from the cue pixels through equal-label 8-neighbours,
// followed by the reference's exclusion rule — integer work, bit-exact.
#include <math.h>
#include "common.h"

namespace dsrg {

constexpr int kSrgWG = 1024;

__global__ __launch_bounds__(kSrgWG) void srg_grow_kernel(int C, int H, int W,
                                                          const float *__restrict__ labels,
                                                          const float *__restrict__ cues,
                                                          const double *__restrict__ refined, double th1,
                                                          double th2, float *__restrict__ seeds) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int N = H * W, Wp = W + 2, Np = (H + 2) * Wp;
    unsigned char *lm = smem;                 // [(H+2)*(W+2)] label-map value (class+1), 0 = none / halo
    unsigned char *grown = smem + ((Np + 15) & ~15);      // same shape: 1 = member of a seeded component
    unsigned char *excl = grown + ((Np + 15) & ~15);      // [N] exclusion rule flag

    const float *lab = labels + (size_t)b * C;
    const float *cu = cues + (size_t)b * C * N;
    const double *rf = refined + (size_t)b * C * N;
    float *out = seeds + (size_t)b * C * N;

    // present classes (labels == 1, pylayers.py:240) as a bit mask; C <= 64
    unsigned long long present = 0ull;
    for (int c = 0; c < C; c++)
        if (lab[c] == 1.0f) present |= 1ull << c;

    for (int p = tid; p < Np; p += kSrgWG) { lm[p] = 0; grown[p] = 0; }
    __syncthreads();

    for (int p = tid; p < N; p += kSrgWG) {
        // label map from the cues: highest cued class wins (pylayers.py:248-250)
        int lmv = 0;
        float cuesum = 0.0f;
        for (int c = 0; c < C; c++) {
            const float s = cu[(size_t)c * N + p];
            if (s > 0.0f) lmv = c + 1;
            cuesum += s;
        }
        // argmax / max over the present classes, first maximum wins (pylayers.py:241-243)
        int best = -1;
        double v = 0.0;
        for (int c = 0; c < C; c++) {
            if ((present >> c) & 1ull) {
                const double t = rf[(size_t)c * N + p];
                if (best < 0 || t > v) { v = t; best = c; }
            }
        }
        if (best >= 0 && v > th2) {            // pylayers.py:253-257, strict float64 compares
            if (best != 0) lmv = best + 1;
            else if (v > th1) lmv = 1;
        }
        const int cls = lmv - 1;
        const bool active = lmv > 0 && ((present >> cls) & 1ull);   // only present classes are grown (:259)
        const float own = active ? cu[(size_t)cls * N + p] : 0.0f;
        const int y = p / W, x = p - y * W;
        const int pp = (y + 1) * Wp + (x + 1);
        lm[pp] = active ? (unsigned char)lmv : 0;
        grown[pp] = (active && own == 1.0f) ? 1 : 0;               // seeds of the component (:266)
        excl[p] = (active && own != 1.0f && cuesum == 1.0f) ? 1 : 0;   // cued by exactly one OTHER class (:268-269)
    }
    __syncthreads();

    // grow to the fixed point: a pixel joins when an 8-neighbour with the same label is a member
    for (;;) {
        int changed = 0;
        for (int p = tid; p < N; p += kSrgWG) {
            const int y = p / W, x = p - y * W;
            const int pp = (y + 1) * Wp + (x + 1);
            const unsigned char l = lm[pp];
            if (l != 0 && grown[pp] == 0) {
                const int up = pp - Wp, dn = pp + Wp;
                const bool hit = (grown[up - 1] && lm[up - 1] == l) || (grown[up] && lm[up] == l) ||
                                 (grown[up + 1] && lm[up + 1] == l) || (grown[pp - 1] && lm[pp - 1] == l) ||
                                 (grown[pp + 1] && lm[pp + 1] == l) || (grown[dn - 1] && lm[dn - 1] == l) ||
                                 (grown[dn] && lm[dn] == l) || (grown[dn + 1] && lm[dn + 1] == l);
                if (hit) { grown[pp] = 1; changed = 1; }
            }
        }
        // frontier empty?  64-bit wave ballot, then OR across the workgroup's waves
        const int wave_changed = __ballot(changed) != 0ull;
        (void)lane;
        if (!__syncthreads_or(wave_changed)) break;
    }

    // seeds = cues, plus every member pixel of its own class unless excluded (pylayers.py:271-273)
    for (int p = tid; p < N; p += kSrgWG) {
        const int y = p / W, x = p - y * W;
        const int pp = (y + 1) * Wp + (x + 1);
        const int cls = (int)lm[pp] - 1;
        const bool add = cls >= 0 && grown[pp] && !excl[p];
        for (int c = 0; c < C; c++) {
            float s = cu[(size_t)c * N + p];
            if (add && c == cls) s = 1.0f;
            out[(size_t)c * N + p] = s;
        }
    }
}

int launch_srg(int B, int C, int H, int W, const float *labels, const float *cues, const double *refined,
               double th1, double th2, float *seeds, hipStream_t stream) {
    if (C < 1 || C > 64) return set_error(DSRG_ERR_UNSUPPORTED, "SRG supports 1..64 classes, got %d", C);
    const size_t Np = (size_t)(H + 2) * (W + 2);
    const size_t lds = 2 * ((Np + 15) & ~(size_t)15) + (size_t)H * W;
    if (lds > 150 * 1024) return set_error(DSRG_ERR_UNSUPPORTED, "SRG map %dx%d exceeds LDS", H, W);
    static size_t granted = 0;
    int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&srg_grow_kernel), lds, granted);
    if (rc) return rc;
    hipLaunchKernelGGL(srg_grow_kernel, dim3(B), dim3(kSrgWG), lds, stream, C, H, W, labels, cues, refined, th1,
                       th2, seeds);
    DSRG_LAUNCH_CHECK();
    return DSRG_OK;
}

}  // namespace dsrg
