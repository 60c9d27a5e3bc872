// C ABI of libdsrg_hip.so (include/dsrg_hip.h): contexts, workspaces, launch sequences.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include <new>
#include "common.h"

namespace dsrg {

static thread_local char g_err[512] = "";
int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace dsrg

using namespace dsrg;

// ---------------------------------------------------------------------------------
struct dsrg_ctx_s {
    int maxB, C, H, W, N;
    void *arena;                 // one hipMalloc for everything below
    LatticeView Lg, Lb;
    LatticeFeat Fg_built;        // parameters the cached Gaussian lattice was built with
    bool gauss_valid;
    MeanfieldBufs mf;
    unsigned char *im_u8;        // (maxB, N, 3)
    // fused-step blobs
    float *probs, *logq, *seeds;
    double *refined;
    double *stats;               // (maxB, 5)
    uint16_t *srg_code;          // (maxB, N) per-pixel codes handed from the SRG classification pass to the growth pass
    Profiler prof;
    int prepared_B;              // batch whose lattices dsrg_crf_prepare_batch built (0 = none)
    dsrg_crf_params prepared_prm;
    int gauss_local;             // the cached Gaussian lattice's kLatticeLocal flag as read back by the host: 1 yes, 0 no,
                                 // -1 not read (built inside a stream capture): the kernels then test the flag themselves
};

namespace dsrg { extern void *g_filter_dbg; extern void *g_build_dbg; extern std::atomic<int> g_filter_opts; extern std::atomic<int> g_igemm_variant; }
// tests / tools only (not in the public header): option bits of the mean-field filter launch (meanfield.hip, kOpt*: 1 = a
// pixel-local Gaussian lattice is evaluated by the update kernel, 2 = slot guard); -1 = back to the DSRG_FILTER_OPTS environment variable / the default (all on).  Every combination yields
// bit-identical marginals (tests/test_gpu_parity.py).
extern "C" __attribute__((visibility("default"))) void dsrg_debug_set_filter_opts(int opts) {
    dsrg::g_filter_opts = opts < 0 ? -1 : (opts & 3);      // bits 4 and 8 (seqCompute arithmetic, norm pass) are the launcher's own
}
// tests / tools only: pipeline of the implicit-GEMM convolution: 1 = two LDS stages of 64 reduction elements, every wave issuing
// its DMA right behind the barrier; 3 (default) = the same with waves 4-7 issuing behind their first MFMA cluster; 2 = a ring of
// four stages of 32 with three steps in flight; 4 = as 3 with the stream-K form wherever it is legal (by default only where it
// wins); 5 = as 3 with the step's barrier in front of its last MFMA cluster and the next step's first fragments read before it; 6 / 7 =
// round 4's launches (every K-step multiplied) / 3 with flat-order skipping in the weight gradient; 8 = 3 with round 5's row-aligned
// pixel tiles for the dilated launches, 9 = 3 with the class-ordered tiles wherever legal (by default only where they run fewer
// K-steps); -1 = back to DSRG_IGEMM_VARIANT / the default; identical results up to the summation order of a cut tile
extern "C" __attribute__((visibility("default"))) void dsrg_debug_set_igemm_variant(int v) { dsrg::g_igemm_variant = v; }
extern "C" __attribute__((visibility("default"))) void dsrg_debug_set_build_trace(void *dev_buf) { dsrg::g_build_dbg = dev_buf; }
// tools only (not in the public header): device buffer of 16 u64 per filter block receiving phase timestamps
extern "C" __attribute__((visibility("default"))) void dsrg_debug_set_filter_trace(void *dev_buf) { dsrg::g_filter_dbg = dev_buf; }

extern "C" const char *dsrg_last_error(void) { return g_err; }

extern "C" int dsrg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

extern "C" int dsrg_host_register(void *host, size_t bytes) {
    if (!host || !bytes) return 0;
    // memory the runtime already knows (its owner page-locked it, or it lies inside an earlier registration) is left alone:
    // registering it again fails AND leaves the failure as the thread's last error for an unrelated launch check to report
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    if (hipPointerGetAttributes(&at, host) == hipSuccess) {
        if (at.type != hipMemoryTypeUnregistered) return 0;
    } else {
        (void)hipGetLastError();                             // (older runtimes answer "invalid value" for plain host memory)
    }
    const char *last = static_cast<const char *>(host) + bytes - 1;
    memset(&at, 0, sizeof(at));
    if (hipPointerGetAttributes(&at, last) == hipSuccess) {
        if (at.type != hipMemoryTypeUnregistered) return 0;
    } else {
        (void)hipGetLastError();
    }
    if (hipHostRegister(host, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return 1;
}

extern "C" int dsrg_host_unregister(void *host) {
    if (!host) return 0;
    if (hipHostUnregister(host) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return 1;
}

extern "C" int dsrg_ctx_create(int max_batch, int C, int H, int W, dsrg_ctx_t *out) {
    if (!out || max_batch < 1 || C < 1 || H < 1 || W < 1) return set_error(DSRG_ERR_INVALID, "bad ctx shape");
    if (C > kMaxLabels) return set_error(DSRG_ERR_UNSUPPORTED, "at most %d labels", kMaxLabels);
    const int N = H * W;
    if (!lattice_supported(2, N) || !lattice_supported(5, N))
        return set_error(DSRG_ERR_UNSUPPORTED,
                         "%dx%d map: %d pixels, the LDS-resident path takes at most %d (a function of the map size alone, "
                         "never of the images); use the object API (dsrg_crf_create / dsrg_crf_create_batch) for such maps",
                         H, W, N, DSRG_CTX_MAX_PIXELS);
    if (dsrg_device_count() < 1) return set_error(DSRG_ERR_HIP, "no HIP device visible");
    dsrg_ctx_s *c = new (std::nothrow) dsrg_ctx_s();
    if (!c) return set_error(DSRG_ERR_NOMEM, "host allocation failed");
    c->maxB = max_batch; c->C = C; c->H = H; c->W = W; c->N = N; c->gauss_valid = false;
    c->prepared_B = 0; c->prof.start = c->prof.stop = nullptr; c->prof.cap = c->prof.used = 0; c->prof.active = false;
    const size_t blob = align256(sizeof(float) * (size_t)max_batch * C * N);
    const size_t szLg = align256(lattice_bytes(2, N, 1)), szLb = align256(lattice_bytes(5, N, max_batch));
    const size_t szIm = align256((size_t)max_batch * N * 3);
    const size_t szRef = align256(sizeof(double) * (size_t)max_batch * C * N);
    const size_t szStats = align256(sizeof(double) * (size_t)max_batch * 5 * 8);   // [B][kStatSplit][5]
    const size_t szCode = align256(sizeof(uint16_t) * (size_t)max_batch * N);
    const size_t total = szLg + szLb + 3 * blob /*mf*/ + szIm + 3 * blob /*probs,logq,seeds*/ + szRef + szStats + szCode + 256;
    hipError_t e = hipMalloc(&c->arena, total);
    if (e != hipSuccess) {
        delete c;
        return set_error(DSRG_ERR_NOMEM, "hipMalloc(%zu) failed: %s", total, hipGetErrorString(e));
    }
    c->gauss_local = -1;
    unsigned char *p = static_cast<unsigned char *>(c->arena);
    lattice_carve(c->Lg, p, 2, N, 1); p += szLg;
    lattice_carve(c->Lb, p, 5, N, max_batch); p += szLb;
    c->mf.q = reinterpret_cast<float *>(p); p += blob;
    c->mf.msg_g = reinterpret_cast<float *>(p); p += blob;
    c->mf.msg_b = reinterpret_cast<float *>(p); p += blob;
    c->im_u8 = p; p += szIm;
    c->probs = reinterpret_cast<float *>(p); p += blob;
    c->logq = reinterpret_cast<float *>(p); p += blob;
    c->seeds = reinterpret_cast<float *>(p); p += blob;
    c->refined = reinterpret_cast<double *>(p); p += szRef;
    c->stats = reinterpret_cast<double *>(p); p += szStats;
    c->srg_code = reinterpret_cast<uint16_t *>(p); p += szCode;
    *out = c;
    return DSRG_OK;
}

static void prof_free(Profiler &p) {
    for (int i = 0; i < p.cap; i++) { (void)hipEventDestroy(p.start[i]); (void)hipEventDestroy(p.stop[i]); }
    delete[] p.start; delete[] p.stop;
    p.start = p.stop = nullptr; p.cap = p.used = 0; p.active = false;
}

static int prof_start(Profiler &p, int max_launches) {
    if (p.cap < max_launches) {
        prof_free(p);
        p.start = new (std::nothrow) hipEvent_t[max_launches];
        p.stop = new (std::nothrow) hipEvent_t[max_launches];
        if (!p.start || !p.stop) return set_error(DSRG_ERR_NOMEM, "host allocation failed");
        for (int i = 0; i < max_launches; i++) {
            DSRG_HIP_CHECK(hipEventCreate(&p.start[i]));
            DSRG_HIP_CHECK(hipEventCreate(&p.stop[i]));
            p.cap = i + 1;
        }
    }
    p.used = 0;
    p.active = true;
    return DSRG_OK;
}
static int prof_stop(Profiler &p, double *total_ms, int32_t *launches) {
    p.active = false;
    double tot = 0.0;
    for (int i = 0; i < p.used; i++) {
        DSRG_HIP_CHECK(hipEventSynchronize(p.stop[i]));
        float ms = 0.f;
        DSRG_HIP_CHECK(hipEventElapsedTime(&ms, p.start[i], p.stop[i]));
        tot += ms;
    }
    *total_ms = tot;
    *launches = p.used;
    return DSRG_OK;
}

extern "C" int dsrg_ctx_profile_start(dsrg_ctx_t c, int max_launches) {
    if (!c || max_launches < 1) return set_error(DSRG_ERR_INVALID, "bad argument");
    return prof_start(c->prof, max_launches);
}

extern "C" int dsrg_ctx_profile_stop(dsrg_ctx_t c, double *total_ms, int32_t *launches) {
    if (!c || !total_ms || !launches) return set_error(DSRG_ERR_INVALID, "bad argument");
    return prof_stop(c->prof, total_ms, launches);
}

extern "C" int dsrg_ctx_destroy(dsrg_ctx_t c) {
    if (!c) return DSRG_OK;
    prof_free(c->prof);
    if (c->arena) (void)hipFree(c->arena);
    delete c;
    return DSRG_OK;
}

static int check_params(const dsrg_crf_params *p) {
    if (!p) return set_error(DSRG_ERR_INVALID, "params is NULL");
    if (p->n_iters < 0) return set_error(DSRG_ERR_INVALID, "n_iters < 0");
    if (!(p->theta_alpha_x > 0 && p->theta_alpha_y > 0 && p->theta_beta_r > 0 && p->theta_beta_g > 0 &&
          p->theta_beta_b > 0 && p->theta_gamma_x > 0 && p->theta_gamma_y > 0))
        return set_error(DSRG_ERR_INVALID, "kernel widths must be positive");
    return DSRG_OK;
}

// the cached Gaussian lattice's kLatticeLocal flag, read back once per (shape, theta_gamma) — one host synchronisation of the
// stream, on the first call after a rebuild that does not run inside a stream capture (documented in dsrg_hip.h)
static int refresh_gauss_local(dsrg_ctx_t c, hipStream_t s) {
    if (c->gauss_local != -1 || !c->gauss_valid) return DSRG_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusActive; }
    if (cap != hipStreamCaptureStatusNone) return DSRG_OK;
    int fl = 0;
    DSRG_HIP_CHECK(hipMemcpyAsync(&fl, c->Lg.flags, sizeof(int), hipMemcpyDeviceToHost, s));
    DSRG_HIP_CHECK(hipStreamSynchronize(s));
    c->gauss_local = (fl & kLatticeLocal) ? 1 : 0;
    return DSRG_OK;
}

// build the lattices for B images (colours: a (B,N,3) uint8 image, or the net's float images resampled on the fly)
static int crf_build(dsrg_ctx_t c, int B, const LatticeColours &col, const dsrg_crf_params *prm, hipStream_t s) {
    LatticeFeat Fg, Fb;
    lattice_feat_init(Fg, 2, c->W, c->H, prm->theta_gamma_x, prm->theta_gamma_y, 1.f, 1.f, 1.f);
    lattice_feat_init(Fb, 5, c->W, c->H, prm->theta_alpha_x, prm->theta_alpha_y, prm->theta_beta_r,
                      prm->theta_beta_g, prm->theta_beta_b);
    int rc;
    // the Gaussian lattice depends only on (W,H,theta_gamma): build once, reuse across calls
    if (!c->gauss_valid || memcmp(&Fg, &c->Fg_built, sizeof(Fg)) != 0) {
        rc = launch_lattice_build(c->Lg, Fg, LatticeColours(), 1, s);
        if (rc) return rc;
        c->Fg_built = Fg;
        c->gauss_valid = true;
        // Once per (shape, theta_gamma): read the lattice's flags back so that later filter launches can leave the Gaussian
        // workgroups out of the grid when the lattice is pixel-local.  Not inside a stream capture (no host sync there):
        // the kernels then decide from the device-side flag and the launch merely carries workgroups that exit at once.
        c->gauss_local = -1;
    }
    // (the flag is read back lazily: here when the stream is not capturing, else by the first later call that is not)
    if (int rc2 = refresh_gauss_local(c, s)) return rc2;
    return launch_lattice_build(c->Lb, Fb, col, B, s);
}

static LatticeColours colours_u8(const unsigned char *im_u8) { LatticeColours c; c.im_u8 = im_u8; return c; }
// the net's mean-subtracted float images, resampled to the map inside the embedding kernel (pylayers.py:70-75); the uint8
// image lands in the context's im_u8 as before
static LatticeColours colours_float(dsrg_ctx_t c, const float *images, int img_h, int img_w) {
    LatticeColours col; col.images = images; col.Hi = img_h; col.Wi = img_w; col.im_out = c->im_u8; return col;
}

// build lattices for B images (unless prepared), then run the mean field
static int crf_run(dsrg_ctx_t c, int B, const float *neg_unary, const LatticeColours &col,
                   const dsrg_crf_params *prm, float *q_out, double *refined, float *logq, hipStream_t s,
                   bool prepared = false, bool q0_ready = false) {
    if (!prepared) {
        int rc = crf_build(c, B, col, prm, s);
        if (rc) return rc;
    } else if (int rc = refresh_gauss_local(c, s)) {        // built inside a capture earlier: learn the flag now
        return rc;
    }
    return launch_meanfield(c->Lg, c->Lb, c->mf, B, c->C, neg_unary, prm->w_gaussian, prm->w_bilateral,
                            prm->n_iters, q_out, refined, logq, c->gauss_local == 1, s, &c->prof, q0_ready);
}

extern "C" int dsrg_crf_prepare_batch(dsrg_ctx_t c, int B, const float *images, int img_h, int img_w,
                                      const dsrg_crf_params *prm, void *stream) {
    if (!c || !images) return set_error(DSRG_ERR_INVALID, "NULL argument");
    if (B < 1 || B > c->maxB) return set_error(DSRG_ERR_INVALID, "batch %d outside 1..%d", B, c->maxB);
    if (img_h < 1 || img_w < 1) return set_error(DSRG_ERR_INVALID, "bad image size");
    int rc = check_params(prm);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = crf_build(c, B, colours_float(c, images, img_h, img_w), prm, s);             // pylayers.py:70-75 inside
    if (rc) return rc;
    c->prepared_B = B;
    c->prepared_prm = *prm;
    return DSRG_OK;
}

extern "C" int dsrg_crf_refine_batch(dsrg_ctx_t c, int B, float *probs, const float *images, int img_h, int img_w,
                                     const dsrg_crf_params *prm, double *refined, float *logq, void *stream) {
    if (!c || !probs || !refined) return set_error(DSRG_ERR_INVALID, "NULL argument");
    if (B < 1 || B > c->maxB) return set_error(DSRG_ERR_INVALID, "batch %d outside 1..%d", B, c->maxB);
    int rc = check_params(prm);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool prepared = images == nullptr;      // lattices were built by dsrg_crf_prepare_batch
    if (prepared) {
        if (c->prepared_B != B || memcmp(&c->prepared_prm, prm, offsetof(dsrg_crf_params, n_iters)) != 0)
            return set_error(DSRG_ERR_INVALID, "images_dev is NULL but dsrg_crf_prepare_batch was not called "
                                               "for this batch size / these kernel parameters");
    } else if (img_h < 1 || img_w < 1) {
        return set_error(DSRG_ERR_INVALID, "bad image size");
    }
    rc = launch_clip_min(probs, (size_t)B * c->C * c->N, s);                    // pylayers.py:67
    if (rc) return rc;
    if (prepared) c->prepared_B = 0;                                              // consumed
    return crf_run(c, B, probs, colours_float(c, images, img_h, img_w), prm, nullptr, refined, logq, s, prepared);    // CRF.py:28: -(-unary) = probs
}

extern "C" int dsrg_crf_meanfield_batch(dsrg_ctx_t c, int B, const float *neg_unary, const unsigned char *im_u8,
                                        const dsrg_crf_params *prm, float *q, void *stream) {
    if (!c || !neg_unary || !im_u8 || !q) return set_error(DSRG_ERR_INVALID, "NULL argument");
    if (B < 1 || B > c->maxB) return set_error(DSRG_ERR_INVALID, "batch %d outside 1..%d", B, c->maxB);
    int rc = check_params(prm);
    if (rc) return rc;
    return crf_run(c, B, neg_unary, colours_u8(im_u8), prm, q, nullptr, nullptr, static_cast<hipStream_t>(stream));
}

extern "C" int dsrg_ctx_lattice_sizes(dsrg_ctx_t c, int B, int32_t *m_gauss, int32_t *m_bil, void *stream) {
    if (!c || B < 0 || B > c->maxB) return set_error(DSRG_ERR_INVALID, "bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (m_gauss) DSRG_HIP_CHECK(hipMemcpyAsync(m_gauss, c->Lg.M, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (m_bil && B > 0)
        DSRG_HIP_CHECK(hipMemcpyAsync(m_bil, c->Lb.M, sizeof(int32_t) * B, hipMemcpyDeviceToHost, s));
    DSRG_HIP_CHECK(hipStreamSynchronize(s));
    return DSRG_OK;
}

// measurement: the number of splat entries beyond the first of their row ("extras", common.h) per lattice — with the vertex
// counts what the LDS traffic model of bench.py needs
extern "C" int dsrg_ctx_lattice_extras(dsrg_ctx_t c, int B, int32_t *x_gauss, int32_t *x_bil, void *stream) {
    if (!c || B < 0 || B > c->maxB) return set_error(DSRG_ERR_INVALID, "bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (x_gauss) DSRG_HIP_CHECK(hipMemcpyAsync(x_gauss, c->Lg.nextra, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (x_bil && B > 0)
        DSRG_HIP_CHECK(hipMemcpyAsync(x_bil, c->Lb.nextra, sizeof(int32_t) * B, hipMemcpyDeviceToHost, s));
    DSRG_HIP_CHECK(hipStreamSynchronize(s));
    return DSRG_OK;
}

extern "C" int dsrg_ctx_filter_plan(dsrg_ctx_t c, int B, int32_t *planes_bilateral, int32_t *planes_gaussian, int32_t *workgroups,
                                    int32_t *lds_bytes) {
    if (!c || B < 1 || B > c->maxB) return set_error(DSRG_ERR_INVALID, "bad argument");
    int out[4];
    int rc = filter_plan_query(c->Lg, c->Lb, c->mf, B, c->C, c->gauss_local == 1, out);
    if (rc) return rc;
    if (planes_bilateral) *planes_bilateral = out[0];
    if (planes_gaussian) *planes_gaussian = out[1];
    if (workgroups) *workgroups = out[2];
    if (lds_bytes) *lds_bytes = out[3];
    return DSRG_OK;
}

// introspection (tests): the lattice of image b (kind 1, bilateral) or the shared Gaussian lattice (kind 0) in the
// reference's own form — keys in id order (HashTable::getKeys, permutohedral.cpp:296), per-pixel vertex ids and
// barycentric weights pixel-major (offset_ / barycentric_, :272-274), blur neighbours per axis with -1 = none (:315-316)
extern "C" int dsrg_ctx_lattice_dump(dsrg_ctx_t c, int kind, int b, int32_t *m_host, int16_t *keys_host, int32_t *vid_host,
                                     float *bary_host, int32_t *n1_host, int32_t *n2_host, void *stream) {
    if (!c || (kind != 0 && kind != 1) || !m_host) return set_error(DSRG_ERR_INVALID, "bad argument");
    const LatticeView &L = kind == 0 ? c->Lg : c->Lb;
    if (b < 0 || b >= L.nlat) return set_error(DSRG_ERR_INVALID, "lattice index %d outside 0..%d", b, L.nlat - 1);
    hipStream_t s = static_cast<hipStream_t>(stream);
    DSRG_HIP_CHECK(hipStreamSynchronize(s));
    const int d = L.d, d1 = d + 1, N = L.N, Mcap = L.Mcap, KW = (d * 16 + 31) / 32;
    int32_t M = 0;
    DSRG_HIP_CHECK(hipMemcpy(&M, L.M + b, sizeof(int32_t), hipMemcpyDeviceToHost));
    *m_host = M;
    if (M < 0 || M > Mcap) return set_error(DSRG_ERR_HIP, "lattice %d has not been built", b);
    if (keys_host) {
        uint32_t *kv = new (std::nothrow) uint32_t[(size_t)(M > 0 ? M : 1) * KW];
        if (!kv) return set_error(DSRG_ERR_NOMEM, "host allocation failed");
        hipError_t e = hipMemcpy(kv, L.key_v + (size_t)b * Mcap * KW, sizeof(uint32_t) * (size_t)M * KW, hipMemcpyDeviceToHost);
        if (e == hipSuccess)
            for (int v = 0; v < M; v++)
                for (int k = 0; k < d; k++)            // embed.h pack_key: (short + 0x8000) in 16-bit fields
                    keys_host[(size_t)v * d + k] =
                        (int16_t)(int)(((kv[(size_t)v * KW + (k >> 1)] >> ((k & 1) * 16)) & 0xFFFFu) - 0x8000u);
        delete[] kv;
        DSRG_HIP_CHECK(e);
    }
    if (vid_host) {
        uint16_t *t = new (std::nothrow) uint16_t[(size_t)d1 * N];
        if (!t) return set_error(DSRG_ERR_NOMEM, "host allocation failed");
        hipError_t e = hipMemcpy(t, L.vid + (size_t)b * d1 * N, sizeof(uint16_t) * (size_t)d1 * N, hipMemcpyDeviceToHost);
        if (e == hipSuccess)
            for (int i = 0; i < N; i++)
                for (int r = 0; r < d1; r++) vid_host[(size_t)i * d1 + r] = t[(size_t)r * N + i];
        delete[] t;
        DSRG_HIP_CHECK(e);
    }
    if (bary_host) {
        float *t = new (std::nothrow) float[(size_t)d1 * N];
        if (!t) return set_error(DSRG_ERR_NOMEM, "host allocation failed");
        hipError_t e = hipMemcpy(t, L.bary + (size_t)b * d1 * N, sizeof(float) * (size_t)d1 * N, hipMemcpyDeviceToHost);
        if (e == hipSuccess)
            for (int i = 0; i < N; i++)
                for (int r = 0; r < d1; r++) bary_host[(size_t)i * d1 + r] = t[(size_t)r * N + i];
        delete[] t;
        DSRG_HIP_CHECK(e);
    }
    if (n1_host || n2_host) {
        uint32_t *t = new (std::nothrow) uint32_t[(size_t)d1 * Mcap];
        if (!t) return set_error(DSRG_ERR_NOMEM, "host allocation failed");
        hipError_t e = hipMemcpy(t, L.nb + (size_t)b * d1 * Mcap, sizeof(uint32_t) * (size_t)d1 * Mcap, hipMemcpyDeviceToHost);
        if (e == hipSuccess)
            for (int j = 0; j < d1; j++)
                for (int v = 0; v < M; v++) {
                    const uint32_t w = t[(size_t)j * Mcap + v];
                    const int a = (int)(w & 0xFFFFu), z = (int)(w >> 16);
                    if (n1_host) n1_host[(size_t)j * M + v] = a == M ? -1 : a;      // slot M = the zero sentinel = none
                    if (n2_host) n2_host[(size_t)j * M + v] = z == M ? -1 : z;
                }
        delete[] t;
        DSRG_HIP_CHECK(e);
    }
    return DSRG_OK;
}

// introspection (tests): the float64 marginals (pylayers.py:84-86, `self.result`) of the last dsrg_supervision_step
extern "C" int dsrg_ctx_read_refined(dsrg_ctx_t c, int B, double *refined_dev, void *stream) {
    if (!c || !refined_dev || B < 1 || B > c->maxB) return set_error(DSRG_ERR_INVALID, "bad argument");
    DSRG_HIP_CHECK(hipMemcpyAsync(refined_dev, c->refined, sizeof(double) * (size_t)B * c->C * c->N, hipMemcpyDeviceToDevice,
                                  static_cast<hipStream_t>(stream)));
    return DSRG_OK;
}

// introspection (tests): norm = 1/sqrt(K 1 + 1e-20) of one lattice (DenseKernel::initLattice, pairwise.cpp:44,54-57)
extern "C" int dsrg_ctx_lattice_norm(dsrg_ctx_t c, int kind, int b, float *norm_host, void *stream) {
    if (!c || (kind != 0 && kind != 1) || !norm_host) return set_error(DSRG_ERR_INVALID, "bad argument");
    const LatticeView &L = kind == 0 ? c->Lg : c->Lb;
    if (b < 0 || b >= L.nlat) return set_error(DSRG_ERR_INVALID, "lattice index %d outside 0..%d", b, L.nlat - 1);
    DSRG_HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    DSRG_HIP_CHECK(hipMemcpy(norm_host, L.norm + (size_t)b * L.N, sizeof(float) * (size_t)L.N, hipMemcpyDeviceToHost));
    return DSRG_OK;
}

// introspection (tests): ONE application of one normalised kernel, out = norm . K (norm . q) (DenseKernel::filter,
// pairwise.cpp:63-80, through Permutohedral::compute, permutohedral.cpp:529-604) with the lattices of the last refine /
// prepare / meanfield / supervision call — exactly the code path the inference loop takes, incl. the per-pixel evaluation
// of a pixel-local Gaussian lattice.  q_dev / out_dev: (B,C,H,W) f32 planes.
extern "C" int dsrg_ctx_filter_once(dsrg_ctx_t c, int kind, int B, const float *q_dev, float *out_dev, void *stream) {
    if (!c || (kind != 0 && kind != 1) || !q_dev || !out_dev) return set_error(DSRG_ERR_INVALID, "bad argument");
    if (B < 1 || B > c->maxB) return set_error(DSRG_ERR_INVALID, "batch %d outside 1..%d", B, c->maxB);
    if (!c->gauss_valid) return set_error(DSRG_ERR_INVALID, "no lattice has been built on this context yet");
    return launch_filter_once(c->Lg, c->Lb, c->mf, B, c->C, kind, q_dev, out_dev, c->gauss_local == 1,
                              static_cast<hipStream_t>(stream));
}

extern "C" int dsrg_crf_layer_backward(size_t n, const double *refined, const float *td, float *bd, void *stream) {
    if (!refined || !td || !bd) return set_error(DSRG_ERR_INVALID, "NULL argument");
    return launch_crf_bwd(n, refined, td, bd, static_cast<hipStream_t>(stream));
}

extern "C" int dsrg_srg_grow_batch(int B, int C, int H, int W, const float *labels, const float *cues,
                                   const double *refined, double th1, double th2, float *seeds, void *scratch, void *stream) {
    if (!labels || !cues || !refined || !seeds || !scratch) return set_error(DSRG_ERR_INVALID, "NULL argument");
    if (B < 1 || H < 1 || W < 1) return set_error(DSRG_ERR_INVALID, "bad shape");
    return launch_srg(B, C, H, W, labels, cues, refined, th1, th2, seeds, static_cast<uint16_t *>(scratch),
                      static_cast<hipStream_t>(stream));
}

extern "C" int dsrg_softmax_forward(int B, int C, int HW, const float *x, float *p, void *stream) {
    if (!x || !p || B < 1 || HW < 1) return set_error(DSRG_ERR_INVALID, "bad argument");
    return launch_softmax_fwd(B, C, HW, x, p, static_cast<hipStream_t>(stream));
}
extern "C" int dsrg_softmax_backward(int B, int C, int HW, const float *x, const float *td, float *bd, void *stream) {
    if (!x || !td || !bd || B < 1 || HW < 1) return set_error(DSRG_ERR_INVALID, "bad argument");
    return launch_softmax_bwd(B, C, HW, x, td, bd, static_cast<hipStream_t>(stream));
}
extern "C" int dsrg_seed_loss(int B, int C, int HW, const float *probs, const float *seeds, float *loss,
                              float *grad, void *stream) {
    if (!probs || !seeds || B < 1 || C < 1 || HW < 1) return set_error(DSRG_ERR_INVALID, "bad argument");
    return launch_seed_loss(B, C, HW, probs, seeds, loss, grad, static_cast<hipStream_t>(stream));
}
extern "C" int dsrg_constrain_loss(int B, int C, int HW, const float *probs, const float *logq, float *loss,
                                   float *gp, float *glq, void *stream) {
    if (!probs || !logq || B < 1 || C < 1 || HW < 1) return set_error(DSRG_ERR_INVALID, "bad argument");
    return launch_constrain_loss(B, C, HW, probs, logq, loss, gp, glq, static_cast<hipStream_t>(stream));
}

extern "C" int dsrg_seed_loss_plain(int B, int C, int HW, const float *probs, const float *seeds, float *loss, float *grad,
                                    void *stream) {
    if (!probs || !seeds || (!loss && !grad)) return set_error(DSRG_ERR_INVALID, "NULL argument");
    if (B <= 0 || C <= 0 || HW <= 0) return set_error(DSRG_ERR_INVALID, "bad shape");
    return launch_seed_loss_plain(B, C, HW, probs, seeds, loss, grad, static_cast<hipStream_t>(stream));
}
extern "C" int dsrg_expand_loss(int B, int C, int HW, const float *probs, const float *stat, double q_fg, double q_bg,
                                float *loss, float *grad, void *scratch, void *stream) {
    if (!probs || !stat || (!loss && !grad)) return set_error(DSRG_ERR_INVALID, "NULL argument");
    if (B <= 0 || C <= 0 || HW <= 0) return set_error(DSRG_ERR_INVALID, "bad shape");
    return launch_expand_loss(B, C, HW, probs, stat, q_fg, q_bg, loss, grad, static_cast<double *>(scratch),
                              static_cast<hipStream_t>(stream));
}
extern "C" int dsrg_confusion_matrix(size_t n, const unsigned char *gt, const unsigned char *pred, int nclass, int rule_lt,
                                     unsigned long long *hist, void *stream) {
    if ((n && (!gt || !pred)) || !hist) return set_error(DSRG_ERR_INVALID, "NULL argument");
    return launch_confusion(n, gt, pred, nclass, rule_lt, hist, static_cast<hipStream_t>(stream));
}
extern "C" int dsrg_im2col3x3_nhwc16(const void *in, void *out, int B, int H, int W, int C, int dilation, void *stream) {
    if (!in || !out || B < 1 || H < 1 || W < 1 || C < 1 || dilation < 1) return set_error(DSRG_ERR_INVALID, "bad argument");
    return launch_im2col3x3(in, out, B, H, W, C, dilation, static_cast<hipStream_t>(stream));
}
extern "C" int dsrg_relu_bwd_bias_bf16(const void *g, const void *y, void *gm, float *bias_grad, float *partials,
                                       int partial_blocks, long rows, int C, float scale, void *stream) {
    if (!g || !bias_grad || !partials || (y && !gm)) return set_error(DSRG_ERR_INVALID, "NULL argument");
    return launch_relu_bwd_bias(g, y, gm, bias_grad, partials, partial_blocks, rows, C, scale, static_cast<hipStream_t>(stream));
}
extern "C" int dsrg_col2im3x3_nhwc_bf16(const void *cols, void *out, int B, int H, int W, int C, int dilation, void *stream) {
    if (!cols || !out || B <= 0 || H <= 0 || W <= 0 || dilation < 1) return set_error(DSRG_ERR_INVALID, "bad col2im arguments");
    return launch_col2im3x3(cols, out, B, H, W, C, dilation, static_cast<hipStream_t>(stream));
}
extern "C" int dsrg_avgpool3x3_s1_bf16(const void *in, void *out, int B, int H, int W, int C, void *stream) {
    if (!in || !out) return set_error(DSRG_ERR_INVALID, "NULL argument");
    return launch_avgpool3x3_s1(in, out, B, H, W, C, static_cast<hipStream_t>(stream));
}
extern "C" int dsrg_conv_igemm_residual_bf16(const void *x_dev, const void *w_dev, const float *bias_dev, const void *res_dev, const void *mask_dev,
                                             void *y_dev, int dilation, int B, int H, int W, int cin, int cout, int ksize, int relu, void *stream) {
    if (!x_dev || !w_dev || !res_dev || !y_dev || B < 1 || H < 1 || W < 1) return set_error(DSRG_ERR_INVALID, "conv_igemm_residual: bad arguments");
    return launch_conv_igemm_residual(x_dev, w_dev, bias_dev, res_dev, mask_dev, y_dev, dilation, B, H, W, cin, cout, ksize, relu,
                                      static_cast<hipStream_t>(stream));
}
extern "C" int dsrg_aspp_shift_sum_f32(const void *y_dev, const float *bias_dev, float *out_dev, const int *offsets, int npairs, int outputs,
                                       int channels, int B, int H, int W, void *stream) {
    if (!y_dev || !out_dev || B < 1 || H < 1 || W < 1) return set_error(DSRG_ERR_INVALID, "aspp_shift_sum: bad arguments");
    return launch_aspp_shift_sum(y_dev, bias_dev, out_dev, offsets, npairs, outputs, channels, B, H, W, static_cast<hipStream_t>(stream));
}
extern "C" int dsrg_aspp_shift_gather_bf16(const float *g_dev, void *gp_dev, const int *offsets, int npairs, int outputs, int channels, int B,
                                           int H, int W, void *stream) {
    if (!g_dev || !gp_dev || B < 1 || H < 1 || W < 1) return set_error(DSRG_ERR_INVALID, "aspp_shift_gather: bad arguments");
    return launch_aspp_shift_gather(g_dev, gp_dev, offsets, npairs, outputs, channels, B, H, W, static_cast<hipStream_t>(stream));
}
extern "C" int dsrg_conv_igemm_split_f32(const void *x3_dev, const void *w_dev, const float *bias_dev, float *y_dev, int dilation, int B, int H,
                                         int W, int cin, int cout, int ksize, int relu, void *stream) {
    if (!x3_dev || !w_dev || !y_dev) return set_error(DSRG_ERR_INVALID, "NULL argument");
    return launch_conv_igemm_split(x3_dev, w_dev, bias_dev, y_dev, dilation, B, H, W, cin, cout, ksize, relu, static_cast<hipStream_t>(stream));
}
extern "C" int dsrg_add_relu_bf16(const void *a, const void *b, void *y, size_t n, void *stream) {
    if (!a || !b || !y) return set_error(DSRG_ERR_INVALID, "NULL argument");
    return launch_add_relu(a, b, y, n, static_cast<hipStream_t>(stream));
}
extern "C" int dsrg_relu_mask_bf16(const void *g, const void *g2, const void *y, void *gm, size_t n, void *stream) {
    if (!g || !y || !gm) return set_error(DSRG_ERR_INVALID, "NULL argument");
    return launch_relu_mask(g, g2, y, gm, n, static_cast<hipStream_t>(stream));
}
extern "C" int dsrg_bias_grad_bf16(const void *g, float *bias_grad, float *partials, int partial_blocks, long rows, int C,
                                   void *stream) {
    if (!g || !bias_grad || !partials) return set_error(DSRG_ERR_INVALID, "NULL argument");
    return launch_bias_grad(g, bias_grad, partials, partial_blocks, rows, C, static_cast<hipStream_t>(stream));
}
extern "C" int dsrg_heads_forward_bf16(const void *const *x_dev, int n_branches, const float *w_dev, const float *bias_dev,
                                       float *out_dev, int B, int HW, int K, int O, void *stream) {
    if (!x_dev || !w_dev || !out_dev || B < 1 || HW < 1) return set_error(DSRG_ERR_INVALID, "bad argument");
    for (int k = 0; k < n_branches && k < 4; k++)
        if (!x_dev[k]) return set_error(DSRG_ERR_INVALID, "NULL branch input");
    return launch_heads_fwd(x_dev, n_branches, w_dev, bias_dev, out_dev, B, HW, K, O, static_cast<hipStream_t>(stream));
}
extern "C" int dsrg_conv3x3_direct_bf16(const void *x_dev, const void *w_dev, const float *bias_dev, void *y_dev, int B, int H,
                                        int W, int cin, int cout, int relu, void *stream) {
    if (!x_dev || !w_dev || !y_dev || B < 1 || H < 1 || W < 1) return set_error(DSRG_ERR_INVALID, "bad argument");
    return launch_conv3x3_direct(x_dev, w_dev, bias_dev, y_dev, B, H, W, cin, cout, relu, static_cast<hipStream_t>(stream));
}
extern "C" size_t dsrg_conv3x3_direct_dgrad_workspace(int cout) { return conv3x3_direct_colsum_workspace(cout); }
extern "C" int dsrg_conv3x3_direct_dgrad_bf16(const void *g_dev, const void *w_dev, const void *mask_dev, void *gx_dev,
                                              float *bias_grad_dev, void *workspace_dev, size_t workspace_bytes, int B, int H, int W,
                                              int cin, int cout, void *stream) {
    if (!g_dev || !w_dev || !mask_dev || !gx_dev || !bias_grad_dev) return set_error(DSRG_ERR_INVALID, "NULL argument");
    return launch_conv3x3_direct(g_dev, w_dev, nullptr, gx_dev, B, H, W, cin, cout, 0, static_cast<hipStream_t>(stream), mask_dev,
                                 bias_grad_dev, workspace_dev, workspace_bytes);
}
extern "C" int dsrg_conv_igemm_supported(int cin, int cout, int ksize) { return conv_igemm_supported(cin, cout, ksize) ? 1 : 0; }
extern "C" int dsrg_conv_igemm_bf16(const void *const *x_dev, const void *const *w_dev, const float *const *bias_dev,
                                    void *const *y_dev, const int *dilation, int ngroups, int B, int H, int W, int cin, int cout,
                                    int ksize, int relu, float dropout_p, unsigned long long dropout_seed, void *workspace_dev,
                                    size_t workspace_bytes, void *stream) {
    if (!x_dev || !w_dev || !y_dev || B < 1 || H < 1 || W < 1) return set_error(DSRG_ERR_INVALID, "conv_igemm: bad arguments");
    return launch_conv_igemm(x_dev, w_dev, bias_dev, y_dev, dilation, ngroups, B, H, W, cin, cout, ksize, relu, dropout_p,
                             dropout_seed, workspace_dev, workspace_bytes, static_cast<hipStream_t>(stream));
}
extern "C" size_t dsrg_conv_igemm_dgrad_workspace(int ngroups, int B, int H, int W, int cout) {
    return conv_igemm_colsum_workspace(ngroups, B, H, W, cout);
}
extern "C" int dsrg_conv_igemm_dgrad_bf16(const void *const *g_dev, const void *const *w_dev, const void *const *mask_dev,
                                          void *const *gx_dev, float *const *bias_grad_dev, const int *dilation, int ngroups, int B,
                                          int H, int W, int cin, int cout, int ksize, float mask_scale, void *workspace_dev,
                                          size_t workspace_bytes, void *stream) {
    if (!g_dev || !w_dev || !gx_dev || !mask_dev || B < 1 || H < 1 || W < 1)
        return set_error(DSRG_ERR_INVALID, "conv_igemm_dgrad: bad arguments");
    return launch_conv_igemm(g_dev, w_dev, nullptr, gx_dev, dilation, ngroups, B, H, W, cin, cout, ksize, 0, 0.0f, 0ull, nullptr, 0,
                             static_cast<hipStream_t>(stream), mask_dev, mask_scale, bias_grad_dev, workspace_dev, workspace_bytes);
}
extern "C" size_t dsrg_conv_igemm_workspace(void) { return conv_igemm_workspace(); }
extern "C" int dsrg_conv_igemm_workspace_status(const void *workspace_dev, void *stream, int *status_host) {
    if (!workspace_dev || !status_host) return set_error(DSRG_ERR_INVALID, "bad argument");
    return conv_igemm_workspace_status(workspace_dev, static_cast<hipStream_t>(stream), status_host);
}
extern "C" int dsrg_pack_conv_weight_f32(const float *w_dev, void *fwd_dev, void *dgrad_dev, int cout, int cin, int ksize, void *stream) {
    return launch_pack_conv_weight(w_dev, fwd_dev, dgrad_dev, cout, cin, ksize, static_cast<hipStream_t>(stream));
}
namespace dsrg {   // sgd_pack.hip
int launch_sgd_pack(int n, float *const *p, const float *const *g, float *const *buf, void *const *fwd, void *const *dg, const int *shape,
                    const long long *numel, const float *lr, const float *wd, float momentum, hipStream_t stream);
}
extern "C" int dsrg_sgd_pack_f32(int n, float *const *param_dev, const float *const *grad_dev, float *const *momentum_dev, void *const *fwd_dev,
                                 void *const *dgrad_dev, const int *shape, const long long *numel, const float *lr, const float *weight_decay,
                                 float momentum, void *stream) {
    if (n > 0 && (!param_dev || !numel || ((fwd_dev || dgrad_dev) && !shape)))
        return set_error(DSRG_ERR_INVALID, "sgd_pack: null argument");
    return launch_sgd_pack(n, param_dev, grad_dev, momentum_dev, fwd_dev, dgrad_dev, shape, numel, lr, weight_decay, momentum,
                           static_cast<hipStream_t>(stream));
}
extern "C" size_t dsrg_conv_igemm_wgrad_workspace(int ngroups, int B, int H, int W, int cin, int cout, int ksize) {
    return conv_igemm_wgrad_workspace(ngroups, B, H, W, cin, cout, ksize);
}
extern "C" int dsrg_conv_igemm_wgrad_bf16(const void *const *x_dev, const void *const *g_dev, void *const *gw_dev, const int *dilation,
                                          int ngroups, void *workspace_dev, size_t workspace_bytes, int B, int H, int W, int cin,
                                          int cout, int ksize, int out_bf16, void *stream) {
    if (!x_dev || !g_dev || !gw_dev || B < 1 || H < 1 || W < 1) return set_error(DSRG_ERR_INVALID, "conv_igemm_wgrad: bad arguments");
    return launch_conv_igemm_wgrad(x_dev, g_dev, gw_dev, dilation, ngroups, workspace_dev, workspace_bytes, B, H, W, cin, cout, ksize,
                                   out_bf16, static_cast<hipStream_t>(stream));
}
namespace dsrg {
int launch_conv_igemm_backward(const void *g, const void *wd, const void *x, const void *mask, void *gx, void *gw, int dil, float *bias_grad,
                               float mask_scale, void *colsum_ws, size_t colsum_ws_bytes, void *wgrad_ws, size_t wgrad_ws_bytes, int B, int H,
                               int W, int cin, int cout, int k, hipStream_t stream);
}
extern "C" int dsrg_conv_igemm_backward_bf16(const void *g_dev, const void *w_dgrad_dev, const void *x_dev, const void *mask_dev, void *gx_dev,
                                             float *gw_dev, int dilation, float *bias_grad_dev, float mask_scale, void *colsum_workspace_dev,
                                             size_t colsum_workspace_bytes, void *wgrad_workspace_dev, size_t wgrad_workspace_bytes, int B,
                                             int H, int W, int cin, int cout, int ksize, void *stream) {
    if (!g_dev || !w_dgrad_dev || !x_dev || !gx_dev || !gw_dev || B < 1 || H < 1 || W < 1 || (bias_grad_dev && !mask_dev))
        return set_error(DSRG_ERR_INVALID, "conv_igemm_backward: bad arguments");
    return dsrg::launch_conv_igemm_backward(g_dev, w_dgrad_dev, x_dev, mask_dev, gx_dev, gw_dev, dilation, bias_grad_dev, mask_scale,
                                            colsum_workspace_dev, colsum_workspace_bytes, wgrad_workspace_dev, wgrad_workspace_bytes, B, H,
                                            W, cin, cout, ksize, static_cast<hipStream_t>(stream));
}
extern "C" int dsrg_conv_igemm_backward_residual_bf16(const void *g_dev, const void *w_dgrad_dev, const void *x_dev, const void *mask_dev,
                                                      const void *res_dev, void *gx_dev, float *gw_dev, const float *gw_scale_dev, int dilation,
                                                      void *wgrad_workspace_dev, size_t wgrad_workspace_bytes, int B, int H, int W, int cin,
                                                      int cout, int ksize, void *stream) {
    if (!g_dev || !w_dgrad_dev || !x_dev || !gx_dev || !gw_dev || B < 1 || H < 1 || W < 1)
        return set_error(DSRG_ERR_INVALID, "conv_igemm_backward_residual: bad arguments");
    return launch_conv_igemm_backward_residual(g_dev, w_dgrad_dev, x_dev, mask_dev, res_dev, gx_dev, gw_dev, gw_scale_dev, dilation,
                                               wgrad_workspace_dev, wgrad_workspace_bytes, B, H, W, cin, cout, ksize,
                                               static_cast<hipStream_t>(stream));
}
extern "C" int dsrg_pack_conv_weight_scaled_f32(const float *w_dev, const float *scale_dev, void *fwd_dev, void *dgrad_dev, int cout, int cin,
                                                int ksize, void *stream) {
    return launch_pack_conv_weight(w_dev, fwd_dev, dgrad_dev, cout, cin, ksize, static_cast<hipStream_t>(stream), 0, scale_dev);
}
extern "C" size_t dsrg_conv3x3_wgrad_workspace(int B, int H, int W, int cin, int cout) {
    return conv3x3_wgrad_workspace(B, H, W, cin, cout);
}
extern "C" int dsrg_conv3x3_wgrad_bf16(const void *x_dev, const void *g_dev, void *gw_dev, void *workspace_dev, size_t workspace_bytes,
                                       int B, int H, int W, int cin, int cout, void *stream) {
    if (!x_dev || !g_dev || !gw_dev || !workspace_dev) return set_error(DSRG_ERR_INVALID, "bad argument");
    return launch_conv3x3_wgrad(x_dev, g_dev, gw_dev, static_cast<float *>(workspace_dev), workspace_bytes, B, H, W, cin, cout,
                                static_cast<hipStream_t>(stream));
}
extern "C" int dsrg_conv3x3_wgrad_f32(const void *x_dev, const void *g_dev, float *gw_dev, void *workspace_dev, size_t workspace_bytes,
                                      int B, int H, int W, int cin, int cout, void *stream) {
    if (!x_dev || !g_dev || !gw_dev || !workspace_dev) return set_error(DSRG_ERR_INVALID, "bad argument");
    return launch_conv3x3_wgrad(x_dev, g_dev, gw_dev, static_cast<float *>(workspace_dev), workspace_bytes, B, H, W, cin, cout,
                                static_cast<hipStream_t>(stream), 1);
}
extern "C" int dsrg_pack_conv_weight_direct_f32(const float *w_dev, void *fwd_dev, void *dgrad_dev, int cout, int cin, void *stream) {
    if ((cin != 64 && cin != 128) || (cout != 64 && cout != 128))
        return set_error(DSRG_ERR_INVALID, "pack_conv_weight_direct: 64 / 128 channels either side (got %d -> %d)", cin, cout);
    return launch_pack_conv_weight(w_dev, fwd_dev, dgrad_dev, cout, cin, 3, static_cast<hipStream_t>(stream), 1);
}
extern "C" int dsrg_heads_backward_chunks(int M) { return heads_bwd_chunks(M); }
extern "C" int dsrg_heads_backward_bf16(const void *const *x_dev, int n_branches, const float *w_dev, const float *g_dev,
                                        void *gx_dev, size_t gx_branch_stride_bytes, float *gw_dev, float *partial_dev, int B,
                                        int HW, int K, int O, void *stream) {
    if (!x_dev || !w_dev || !g_dev || (!gx_dev && !gw_dev) || (gw_dev && !partial_dev) || B < 1 || HW < 1)
        return set_error(DSRG_ERR_INVALID, "bad argument");
    for (int k = 0; k < n_branches && k < 4; k++)
        if (!x_dev[k]) return set_error(DSRG_ERR_INVALID, "NULL branch input");
    return launch_heads_bwd(x_dev, n_branches, w_dev, g_dev, gx_dev, gx_branch_stride_bytes, gw_dev, partial_dev, B, HW, K, O,
                            static_cast<hipStream_t>(stream));
}
extern "C" size_t dsrg_heads_backward_relu_workspace(int n_branches, int M, int K) { return heads_bwd_relu_workspace(n_branches, M, K); }
extern "C" int dsrg_heads_backward_relu_bf16(const void *const *x_dev, int n_branches, const float *w_dev, const float *g_dev,
                                             void *gx_dev, size_t gx_branch_stride_bytes, float *gw_dev, float *partial_dev,
                                             float relu_scale, float *bias_grad_dev, void *workspace_dev, size_t workspace_bytes,
                                             int B, int HW, int K, int O, void *stream) {
    if (!x_dev || !w_dev || !g_dev || !gx_dev || (gw_dev && !partial_dev) || B < 1 || HW < 1 || !(relu_scale > 0.0f) || !bias_grad_dev)
        return set_error(DSRG_ERR_INVALID, "bad argument");
    for (int k = 0; k < n_branches && k < 4; k++)
        if (!x_dev[k]) return set_error(DSRG_ERR_INVALID, "NULL branch input");
    return launch_heads_bwd(x_dev, n_branches, w_dev, g_dev, gx_dev, gx_branch_stride_bytes, gw_dev, partial_dev, B, HW, K, O,
                            static_cast<hipStream_t>(stream), relu_scale, bias_grad_dev, workspace_dev, workspace_bytes);
}
extern "C" int dsrg_maxpool3x3_fwd_bf16(const void *in, void *out, void *code, int B, int H, int W, int OH, int OW, int C,
                                        int stride, void *stream) {
    if (!in || !out || !code) return set_error(DSRG_ERR_INVALID, "NULL argument");
    return launch_maxpool3x3_fwd(in, out, code, B, H, W, OH, OW, C, stride, static_cast<hipStream_t>(stream));
}
extern "C" int dsrg_maxpool3x3_relu_fwd_bf16(const void *in, void *out, void *code, int B, int H, int W, int OH, int OW, int C,
                                             int stride, void *stream) {
    if (!in || !out || !code) return set_error(DSRG_ERR_INVALID, "NULL argument");
    return launch_maxpool3x3_fwd(in, out, code, B, H, W, OH, OW, C, stride, static_cast<hipStream_t>(stream), true);
}
extern "C" int dsrg_maxpool3x3_bwd_bf16(const void *gout, const void *code, void *gin, int B, int H, int W, int OH, int OW,
                                        int C, int stride, void *stream) {
    if (!gout || !code || !gin) return set_error(DSRG_ERR_INVALID, "NULL argument");
    return launch_maxpool3x3_bwd(gout, code, gin, B, H, W, OH, OW, C, stride, static_cast<hipStream_t>(stream));
}

extern "C" int dsrg_maxpool3x3_bwd_relu_bf16(const void *gout, const void *code, const void *relu_out, void *gin, float *bias_grad,
                                             float *partials, int partial_blocks, int B, int H, int W, int OH, int OW, int C,
                                             void *stream) {
    if (!gout || !code || !gin || !bias_grad || !partials) return set_error(DSRG_ERR_INVALID, "NULL argument");
    return launch_maxpool3x3_bwd_relu(gout, code, relu_out, gin, bias_grad, partials, partial_blocks, B, H, W, OH, OW, C,
                                      static_cast<hipStream_t>(stream));
}

extern "C" int dsrg_supervision_step(dsrg_ctx_t c, int B, const float *logits, const float *images, int img_h,
                                     int img_w, const float *labels, const float *cues, double th1, double th2,
                                     const dsrg_crf_params *prm, float *losses, float *grad_logits,
                                     float *probs_out, float *seeds_out, float *logq_out, void *stream) {
    if (!c || !logits || !labels || !cues || !losses || !grad_logits)
        return set_error(DSRG_ERR_INVALID, "NULL argument");
    if (B < 1 || B > c->maxB) return set_error(DSRG_ERR_INVALID, "batch %d outside 1..%d", B, c->maxB);
    const bool prepared = images == nullptr;      // lattices were built by dsrg_crf_prepare_batch
    {
        int prc = check_params(prm);
        if (prc) return prc;
    }
    if (!prepared && (img_h < 1 || img_w < 1)) return set_error(DSRG_ERR_INVALID, "bad image size");
    if (prepared) {
        if (c->prepared_B != B || memcmp(&c->prepared_prm, prm, offsetof(dsrg_crf_params, n_iters)) != 0)
            return set_error(DSRG_ERR_INVALID, "images_dev is NULL but dsrg_crf_prepare_batch was not called "
                                               "for this batch size / these kernel parameters");
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int C = c->C, N = c->N;
    const size_t nb = sizeof(float) * (size_t)B * C * N;
    // Softmax, with the in-place clip CRFLayer.forward applies to this blob next (pylayers.py:67) folded into the same pass:
    // the unclipped blob has no other reader (A.3: both losses read it after the CRF layer ran)
    // ... and with the mean field's starting point Q0 = expAndNormalize(probs) (the unary energy is -probs, CRF.py:28)
    const bool q0 = prm->n_iters > 0;
    int rc = launch_softmax_fwd(B, C, N, logits, c->probs, s, kMinProb, q0 ? c->mf.q : nullptr);
    if (rc) return rc;
    // CRF (once); the images are resampled (pylayers.py:70-75) by the lattices' embedding kernel
    rc = crf_run(c, B, c->probs, colours_float(c, images, img_h, img_w), prm, nullptr, c->refined, c->logq, s, prepared, q0);
    if (prepared) c->prepared_B = 0;                                               // consumed
    if (rc) return rc;
    rc = launch_srg(B, C, c->H, c->W, labels, cues, c->refined, th1, th2, c->seeds, c->srg_code, s);    // DSRG
    if (rc) return rc;
    rc = launch_sup_loss_backward(B, C, N, logits, c->probs, c->seeds, c->logq, c->refined, c->stats, grad_logits,
                                  losses, s);                                        // losses + backward
    if (rc) return rc;
    if (probs_out) DSRG_HIP_CHECK(hipMemcpyAsync(probs_out, c->probs, nb, hipMemcpyDeviceToDevice, s));
    if (seeds_out) DSRG_HIP_CHECK(hipMemcpyAsync(seeds_out, c->seeds, nb, hipMemcpyDeviceToDevice, s));
    if (logq_out) DSRG_HIP_CHECK(hipMemcpyAsync(logq_out, c->logq, nb, hipMemcpyDeviceToDevice, s));
    return DSRG_OK;
}

namespace dsrg {
struct LargeCrf;
int large_crf_create(int W, int H, int C, LargeCrf **out, int nimages);
void large_crf_destroy(LargeCrf *c);
int large_crf_set_unary(LargeCrf *c, const float *unary_host);
int large_crf_zero_unary(LargeCrf *c);
int large_crf_set_image(LargeCrf *c, const unsigned char *im_host);
int large_crf_infer(LargeCrf *c, const dsrg_crf_params *prm, int n_iters);
int large_crf_read_q(LargeCrf *c, float *out_host);
int large_crf_read_map(LargeCrf *c, int32_t *labels_host);
int large_crf_lattice_size(LargeCrf *c, int k);
Profiler *large_crf_profiler(LargeCrf *c);
void large_crf_set_stream(LargeCrf *c, hipStream_t s, bool async);
}  // namespace dsrg

// ---------------------------------------------------------------------------------
// single-image object (DenseCRFWrapper): host pointers, synchronous.  Maps that fit the LDS-resident
// kernels (training sizes) use the batched context with B = 1; larger maps (test-time CRF at image
// resolution) use the global-memory path of lattice_large.hip.
struct dsrg_crf_s {
    int W, H, M;
    int nimg;                    // images per call: 1, or the batch size of dsrg_crf_create_batch (global-memory path only)
    bool forced_large;           // made by dsrg_crf_create_batch: on the global-memory path whatever the map size
    dsrg::LargeCrf *large;
    dsrg_ctx_t ctx;
    float *neg_unary;            // device (M,N) planes = -U
    float *q;                    // device (M,N)
    float *stage;                // device N*M label-fastest staging
    unsigned char *im;           // device N*3
    int32_t *lab;                // device N
    bool have_unary, have_pairwise;
    dsrg_crf_params prm;
    hipStream_t stream;          // dsrg_crf_set_stream: where this object's copies and kernels run (default: the null stream)
    bool async;                  // ... and whether its entry points return without waiting for them
};

// krahenbuhl2013.CRF() creates and destroys one object per call (CRF.py:25): destroyed objects are
// parked (device buffers and all) and handed out again for the same shape, so the per-call cost is the
// inference, not hipMalloc/hipFree of a few hundred MB.
static constexpr int kCrfCacheSlots = 4;
static dsrg_crf_s *g_crf_cache[kCrfCacheSlots] = {nullptr, nullptr, nullptr, nullptr};
static std::mutex g_crf_cache_mutex;

static int crf_create(int W, int H, int nlabels, int nimages, dsrg_crf_t *out);
extern "C" int dsrg_crf_create(int W, int H, int nlabels, dsrg_crf_t *out) { return crf_create(W, H, nlabels, 1, out); }
// nimages same-sized images per call (training/tools/test-ms.py:84-111 and generate_train_gt.py:78-106 loop over 10 582 images
// one at a time: here `nimages` of them share every launch of the build and of the mean-field loop).  Always the global-memory
// path; unary / image / result buffers hold the images back to back; 1 <= nimages <= 8.
extern "C" int dsrg_crf_create_batch(int W, int H, int nlabels, int nimages, dsrg_crf_t *out) {
    if (nimages < 1) return set_error(DSRG_ERR_INVALID, "bad CRF batch");
    return crf_create(W, H, nlabels, nimages == 1 ? -1 : nimages, out);          // (-1: one image, but on the batched objects' path)
}
static int crf_create(int W, int H, int nlabels, int nimages, dsrg_crf_t *out) {
    const bool force_large = nimages != 1;
    if (nimages < 0) nimages = 1;
    if (!out || W < 1 || H < 1 || nlabels < 1) return set_error(DSRG_ERR_INVALID, "bad CRF shape");
    {
        std::lock_guard<std::mutex> lock(g_crf_cache_mutex);
        for (int i = 0; i < kCrfCacheSlots; i++) {
            dsrg_crf_s *c = g_crf_cache[i];
            // (an object is handed back only to the entry point that made it: a one-image object of dsrg_crf_create_batch lives on
            // the global-memory path also for a map the LDS path takes, and dsrg_crf_create must not get that other implementation)
            if (c && c->W == W && c->H == H && c->M == nlabels && c->nimg == nimages && c->forced_large == force_large) {
                g_crf_cache[i] = nullptr;
                c->have_unary = c->have_pairwise = false;
                c->stream = nullptr; c->async = false;
                if (c->large) large_crf_set_stream(c->large, nullptr, false);
                *out = c;
                return DSRG_OK;
            }
        }
    }
    dsrg_crf_s *h = new (std::nothrow) dsrg_crf_s();
    if (!h) return set_error(DSRG_ERR_NOMEM, "host allocation failed");
    memset(h, 0, sizeof(*h));
    h->W = W; h->H = H; h->M = nlabels; h->nimg = nimages; h->forced_large = force_large;
    if (nlabels > kMaxLabels) { delete h; return set_error(DSRG_ERR_UNSUPPORTED, "at most %d labels", kMaxLabels); }
    if (dsrg_device_count() < 1) { delete h; return set_error(DSRG_ERR_HIP, "no HIP device visible"); }
    if (force_large || !lattice_supported(2, W * H) || !lattice_supported(5, W * H)) {
        int rc = large_crf_create(W, H, nlabels, &h->large, nimages);
        if (rc) { delete h; return rc; }
        *out = h;
        return DSRG_OK;
    }
    int rc = dsrg_ctx_create(1, nlabels, H, W, &h->ctx);
    if (rc) { delete h; return rc; }
    const size_t n = (size_t)W * H * nlabels;
    hipError_t e = hipMalloc(&h->neg_unary, sizeof(float) * n);
    if (e == hipSuccess) e = hipMalloc(&h->q, sizeof(float) * n);
    if (e == hipSuccess) e = hipMalloc(&h->stage, sizeof(float) * n);
    if (e == hipSuccess) e = hipMalloc(&h->im, (size_t)W * H * 3);
    if (e == hipSuccess) e = hipMalloc(&h->lab, sizeof(int32_t) * (size_t)W * H);
    if (e != hipSuccess) { dsrg_crf_destroy(h); return set_error(DSRG_ERR_NOMEM, "hipMalloc: %s", hipGetErrorString(e)); }
    *out = h;
    return DSRG_OK;
}
static int crf_free(dsrg_crf_t h);
extern "C" int dsrg_crf_destroy(dsrg_crf_t h) {
    if (!h) return DSRG_OK;
    dsrg_crf_s *evict = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_crf_cache_mutex);
        int slot = -1;
        for (int i = 0; i < kCrfCacheSlots; i++) if (!g_crf_cache[i]) { slot = i; break; }
        if (slot < 0) { evict = g_crf_cache[0]; for (int i = 0; i + 1 < kCrfCacheSlots; i++) g_crf_cache[i] = g_crf_cache[i + 1]; slot = kCrfCacheSlots - 1; }
        g_crf_cache[slot] = h;
    }
    return evict ? crf_free(evict) : DSRG_OK;
}
static int crf_free(dsrg_crf_t h) {
    if (h->large) { large_crf_destroy(h->large); delete h; return DSRG_OK; }
    if (h->neg_unary) (void)hipFree(h->neg_unary);
    if (h->q) (void)hipFree(h->q);
    if (h->stage) (void)hipFree(h->stage);
    if (h->im) (void)hipFree(h->im);
    if (h->lab) (void)hipFree(h->lab);
    dsrg_ctx_destroy(h->ctx);
    delete h;
    return DSRG_OK;
}
extern "C" int dsrg_crf_npixels(dsrg_crf_t h) { return h ? h->W * h->H * h->nimg : 0; }     // of all images of a batched object
extern "C" int dsrg_crf_nlabels(dsrg_crf_t h) { return h ? h->M : 0; }

extern "C" int dsrg_crf_set_unary_energy(dsrg_crf_t h, const float *unary_host) {
    if (!h || !unary_host) return set_error(DSRG_ERR_INVALID, "NULL argument");
    if (h->large) { int rc = large_crf_set_unary(h->large, unary_host); if (!rc) h->have_unary = true; return rc; }
    const int N = h->W * h->H;
    DSRG_HIP_CHECK(hipMemcpyAsync(h->stage, unary_host, sizeof(float) * (size_t)N * h->M, hipMemcpyDefault, h->stream));
    int rc = launch_lf_to_planes(N, h->M, h->stage, h->neg_unary, 1, h->stream);   // inference uses -unary (densecrf.cpp:120,122)
    if (rc) return rc;
    if (!h->async) DSRG_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->have_unary = true;
    return DSRG_OK;
}
extern "C" int dsrg_crf_add_pairwise_energy(dsrg_crf_t h, float w1, float ta1, float ta2, float tb1, float tb2,
                                            float tb3, float w2, float tg1, float tg2, const unsigned char *im_host) {
    if (!h || !im_host) return set_error(DSRG_ERR_INVALID, "NULL argument");
    dsrg_crf_params p;
    p.w_bilateral = w1; p.theta_alpha_x = ta1; p.theta_alpha_y = ta2;
    p.theta_beta_r = tb1; p.theta_beta_g = tb2; p.theta_beta_b = tb3;
    p.w_gaussian = w2; p.theta_gamma_x = tg1; p.theta_gamma_y = tg2; p.n_iters = 0;
    int rc = check_params(&p);
    if (rc) return rc;
    if (h->large) { rc = large_crf_set_image(h->large, im_host); if (rc) return rc; }
    else {
        DSRG_HIP_CHECK(hipMemcpyAsync(h->im, im_host, (size_t)h->W * h->H * 3, hipMemcpyDefault, h->stream));
        if (!h->async) DSRG_HIP_CHECK(hipStreamSynchronize(h->stream));
    }
    h->prm = p;
    h->have_pairwise = true;
    return DSRG_OK;
}
static int crf_infer(dsrg_crf_t h, int n_iters) {
    if (n_iters < 0) return set_error(DSRG_ERR_INVALID, "n_iters < 0");
    const int N = h->W * h->H;
    if (!h->have_pairwise)
        return set_error(DSRG_ERR_INVALID, "add_pairwise_energy must be called before inference");
    dsrg_crf_params p = h->prm;
    p.n_iters = n_iters;
    if (h->large) {
        if (!h->have_unary) { int rc = large_crf_zero_unary(h->large); if (rc) return rc; h->have_unary = true; }
        return large_crf_infer(h->large, &p, n_iters);
    }
    if (!h->have_unary) {    // DenseCRF::inference starts from a zero unary when none was set (densecrf.cpp:117-119)
        DSRG_HIP_CHECK(hipMemsetAsync(h->neg_unary, 0, sizeof(float) * (size_t)N * h->M, h->stream));
        h->have_unary = true;
    }
    return dsrg_crf_meanfield_batch(h->ctx, 1, h->neg_unary, h->im, &p, h->q, h->stream);
}
extern "C" int dsrg_crf_inference(dsrg_crf_t h, int n_iters, float *out_host) {
    if (!h || !out_host) return set_error(DSRG_ERR_INVALID, "NULL argument");
    int rc = crf_infer(h, n_iters);
    if (rc) return rc;
    if (h->large) return large_crf_read_q(h->large, out_host);
    const int N = h->W * h->H;
    rc = launch_planes_to_lf(N, h->M, h->q, h->stage, h->stream);
    if (rc) return rc;
    DSRG_HIP_CHECK(hipMemcpyAsync(out_host, h->stage, sizeof(float) * (size_t)N * h->M, hipMemcpyDefault, h->stream));
    // this API returns only when `out` is final — unless the caller asked for asynchronous calls (dsrg_crf_set_stream)
    if (!h->async) DSRG_HIP_CHECK(hipStreamSynchronize(h->stream));
    return DSRG_OK;
}
extern "C" int dsrg_crf_map(dsrg_crf_t h, int n_iters, int32_t *labels_host) {
    if (!h || !labels_host) return set_error(DSRG_ERR_INVALID, "NULL argument");
    int rc = crf_infer(h, n_iters);
    if (rc) return rc;
    if (h->large) return large_crf_read_map(h->large, labels_host);
    const int N = h->W * h->H;
    rc = launch_argmax_planes(N, h->M, h->q, h->lab, h->stream);
    if (rc) return rc;
    DSRG_HIP_CHECK(hipMemcpyAsync(labels_host, h->lab, sizeof(int32_t) * (size_t)N, hipMemcpyDefault, h->stream));
    if (!h->async) DSRG_HIP_CHECK(hipStreamSynchronize(h->stream));      // see dsrg_crf_inference
    return DSRG_OK;
}
// Where an object's work runs: `stream` (NULL = the null stream), and with async != 0 its entry points enqueue and return —
// pointers passed in must then be device (or pinned) memory that stays valid, and results are final after
// dsrg_crf_synchronize.  Several objects on several streams overlap their (launch-bound) kernels: the test-time loop over
// 10 582 images (training/tools/test-ms.py:84-111) keeps a few images in flight this way.
extern "C" int dsrg_crf_set_stream(dsrg_crf_t h, void *stream, int async) {
    if (!h) return set_error(DSRG_ERR_INVALID, "NULL handle");
    h->stream = static_cast<hipStream_t>(stream);
    h->async = async != 0;
    if (h->large) large_crf_set_stream(h->large, h->stream, h->async);
    return DSRG_OK;
}
extern "C" int dsrg_crf_synchronize(dsrg_crf_t h) {
    if (!h) return set_error(DSRG_ERR_INVALID, "NULL handle");
    DSRG_HIP_CHECK(hipStreamSynchronize(h->stream));
    return DSRG_OK;
}
// measurement hook of the object API: brackets the dominant kernel of this handle's path with HIP events on its stream —
// lg_splat2_kernel (one launch per iteration) on the global-memory path, the mean-field filter kernel on the LDS-resident path
extern "C" int dsrg_crf_profile_start(dsrg_crf_t h, int max_launches) {
    if (!h || max_launches < 1) return set_error(DSRG_ERR_INVALID, "bad argument");
    return prof_start(h->large ? *large_crf_profiler(h->large) : h->ctx->prof, max_launches);
}
extern "C" int dsrg_crf_profile_stop(dsrg_crf_t h, double *total_ms, int32_t *launches) {
    if (!h || !total_ms || !launches) return set_error(DSRG_ERR_INVALID, "bad argument");
    return prof_stop(h->large ? *large_crf_profiler(h->large) : h->ctx->prof, total_ms, launches);
}
extern "C" int dsrg_crf_lattice_size(dsrg_crf_t h, int k) {
    if (h && h->large && k >= 0 && k <= 1) return large_crf_lattice_size(h->large, k);
    if (!h || !h->ctx || k < 0 || k > 1) return -1;
    int32_t m = -1;
    if (hipMemcpy(&m, k == 0 ? h->ctx->Lg.M : h->ctx->Lb.M, sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return m;
}
