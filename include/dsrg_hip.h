/*
 * dsrg_hip.h — C ABI of libdsrg_hip.so, the MI355X (gfx950) implementation of
 * the DSRG per-iteration supervision path.
 *
 * Every entry point replaces one interface of the reference (speedinghzl/DSRG);
 * the reference file:line it stands in for is cited on each declaration.  The
 * library has no CPU fallback: every compute entry point launches HIP kernels
 * and returns an error code when no device / too large a problem is given.
 *
 * Conventions
 *   - return value: 0 = DSRG_OK, negative = error (dsrg_last_error() has text)
 *   - "dev" pointers are device (HBM) pointers owned by the caller; "host"
 *     pointers are ordinary host memory
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); device
 *     entry points are stream-ordered and never synchronise the host
 *   - layouts: blobs are NCHW float32, C-contiguous, exactly like the Caffe
 *     blobs the reference's Python layers see (pylayers.py)
 *   - threads: a handle (dsrg_ctx_t, dsrg_crf_t) is used by one host thread at a
 *     time; different handles, and the handle-free entry points, may be called
 *     from different host threads concurrently (the library's shared tables are
 *     atomics; dsrg_last_error() is per thread).  Results do not depend on the
 *     interleaving (tests/test_gpu_parity.py::test_four_host_threads_four_contexts)
 */
#ifndef DSRG_HIP_H
#define DSRG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#pragma GCC visibility push(default)

#define DSRG_OK 0
#define DSRG_ERR_INVALID (-1)      /* bad argument */
#define DSRG_ERR_HIP (-2)          /* HIP runtime error (no device, launch failure, ...) */
#define DSRG_ERR_UNSUPPORTED (-3)  /* problem does not fit the implemented kernels */
#define DSRG_ERR_NOMEM (-4)

const char *dsrg_last_error(void);
/* number of visible HIP devices (0 when there is none); never fails */
int dsrg_device_count(void);
/* Page-lock `bytes` of ordinary host memory at `host` in place (hipHostRegister), so that the copies between a Caffe blob
 * (pylayers.py: bottom[i].data / top[i].data are host arrays the framework owns) and HBM run as DMA instead of through a
 * bounce buffer.  Returns 1 = registered by this call, 0 = left alone: memory HIP already knows (page-locked by its owner —
 * a hipHostMalloc'ed blob of a GPU-mode Caffe, a pinned torch tensor — or overlapping an earlier registration) or a refused
 * registration.  Never fails and never leaves a HIP error behind for the next launch check to trip over.
 * dsrg_host_unregister undoes a registration made here (call it only for pointers that returned 1); same guarantees. */
int dsrg_host_register(void *host, size_t bytes);
int dsrg_host_unregister(void *host);

/* ------------------------------------------------------------------------ */
/* 1. Single-image dense-CRF object: mirrors the C++ class DenseCRFWrapper     */
/*    (CRF/include/densecrf_wrapper.h:3-28) that the Cython class DenseCRF     */
/*    (CRF/krahenbuhl2013/wrapper.pyx:5-17,20-60) binds.  Host pointers, one   */
/*    image, synchronous — exactly the reference's calling convention.         */
/* ------------------------------------------------------------------------ */
typedef struct dsrg_crf_s *dsrg_crf_t;

/* DenseCRFWrapper::DenseCRFWrapper(int W, int H, int nlabels)  densecrf_wrapper.cpp:5-8 */
int dsrg_crf_create(int W, int H, int nlabels, dsrg_crf_t *out);
/* The same object for `nimages` (1 .. 8) same-sized images per call: what the reference's test-time loops do one image at
 * a time (training/tools/test-ms.py:84-111, generate_train_gt.py:78-106: CRF() per image over 10 582 images) as ONE set of
 * launches — the lattices of the batch are built together (every key carries its image's number, so images share no vertex)
 * and every splat / blur / slice launch of the mean-field loop carries all of them.  The unary, image, result and label
 * buffers of the calls below then hold the images back to back ((nimages, H, W, nlabels) / (nimages, H, W, 3) / (nimages, H,
 * W)); dsrg_crf_npixels returns nimages * W * H.  Results equal nimages single-image calls bit for bit.  Always the
 * global-memory path; DSRG_ERR_UNSUPPORTED at inference when a spatial kernel is so narrow that the image number does not fit
 * beside its lattice coordinates (theta_gamma below ~0.3 pixel at 500 pixels a side): filter such images one at a time. */
int dsrg_crf_create_batch(int W, int H, int nlabels, int nimages, dsrg_crf_t *out);
/* DenseCRFWrapper::~DenseCRFWrapper                             densecrf_wrapper.cpp:10-12 */
int dsrg_crf_destroy(dsrg_crf_t h);
/* The four data pointers of this object API (`unary_host`, `im_host`, `out_host`, `labels_host`) may be host OR device
 * pointers — the copies use hipMemcpyDefault — so a device-resident caller (dsrg_amd/inference.py) skips the PCIe trips.
 * DenseCRFWrapper::set_unary_energy(float*)                     densecrf_wrapper.cpp:32-37
 * unary_host: [W*H*nlabels], label-fastest (copied). */
int dsrg_crf_set_unary_energy(dsrg_crf_t h, const float *unary_host);
/* DenseCRFWrapper::add_pairwise_energy(...)                     densecrf_wrapper.cpp:19-30
 * Gaussian potential (w2, theta_gamma) first, then bilateral (w1, theta_alpha,
 * theta_beta), both Potts.  im_host: [W*H*3] uint8, not retained. */
int dsrg_crf_add_pairwise_energy(dsrg_crf_t h, float w1, float theta_alpha_1, float theta_alpha_2,
                                 float theta_beta_1, float theta_beta_2, float theta_beta_3,
                                 float w2, float theta_gamma_1, float theta_gamma_2,
                                 const unsigned char *im_host);
/* DenseCRFWrapper::inference(int, float*)                       densecrf_wrapper.cpp:45-50
 * out_host: [W*H*nlabels] label-fastest marginals. */
int dsrg_crf_inference(dsrg_crf_t h, int n_iters, float *out_host);
/* DenseCRFWrapper::map(int, int*)                               densecrf_wrapper.cpp:39-43 */
int dsrg_crf_map(dsrg_crf_t h, int n_iters, int32_t *labels_host);
/* Where an object's copies and kernels run (default: the null stream, every call synchronous as the reference's).  With
 * async != 0 the entry points enqueue and return: buffers passed in must be device (or pinned host) memory that stays valid,
 * results are final after dsrg_crf_synchronize.  Objects on different streams overlap their launches — how a test-time loop
 * (training/tools/test-ms.py:84-111) keeps several images in flight (dsrg_amd.crf.CRF_device_many). */
int dsrg_crf_set_stream(dsrg_crf_t crf, void *stream, int async);
int dsrg_crf_synchronize(dsrg_crf_t crf);
/* DenseCRFWrapper::npixels / nlabels                            densecrf_wrapper.cpp:14-15 */
int dsrg_crf_npixels(dsrg_crf_t h);
int dsrg_crf_nlabels(dsrg_crf_t h);
/* introspection (tests): vertex count of lattice k (0 Gaussian, 1 bilateral), -1 if not built */
int dsrg_crf_lattice_size(dsrg_crf_t h, int k);

/* measurement hook (no reference counterpart): while on, every launch of the dominant kernel of this object's path — the
 * splat of the global-memory path (full-resolution maps), the mean-field filter kernel of the LDS-resident path — is
 * bracketed by HIP events on its stream; _stop synchronises and returns the summed time and the launch count. */
int dsrg_crf_profile_start(dsrg_crf_t h, int max_launches);
int dsrg_crf_profile_stop(dsrg_crf_t h, double *total_ms_host, int32_t *launches_host);

/* ------------------------------------------------------------------------ */
/* 2. Batched device context: workspace for B images of (C,H,W) so that the    */
/*    per-iteration path runs without allocation or host synchronisation.      */
/* ------------------------------------------------------------------------ */
typedef struct dsrg_ctx_s *dsrg_ctx_t;

/* pairwise parameters of krahenbuhl2013.CRF (CRF/krahenbuhl2013/CRF.py:25-35);
 * the caller evaluates 80/scale_factor etc. in double and rounds to float, as
 * the Python->Cython call does. */
typedef struct {
    float w_bilateral;      /* 10 */
    float theta_alpha_x;    /* 80 / scale_factor */
    float theta_alpha_y;
    float theta_beta_r;     /* color_factor = 13 */
    float theta_beta_g;
    float theta_beta_b;
    float w_gaussian;       /* 3 */
    float theta_gamma_x;    /* 3 / scale_factor */
    float theta_gamma_y;
    int32_t n_iters;        /* maxiter = 10 */
} dsrg_crf_params;

/* Which maps a context takes is a function of N = H * W alone, never of the images: every lattice is provisioned for its worst
 * case (6 * 4 * ceil(N / 4) vertices for the bilateral kernel, whatever the colours), so that a training run cannot meet a
 * batch that does not fit.  Accepted: N <= DSRG_CTX_MAX_PIXELS (41x41, 65x65 — the maps of 321x321 and 513x513 inputs — up
 * to 66x68; 67x67 is the first square map beyond), C <= 96 labels; anything larger returns DSRG_ERR_UNSUPPORTED here, at
 * creation, with the reason in dsrg_last_error().  Larger maps run on the global-memory path: the object API (section 1,
 * dsrg_crf_create / dsrg_crf_create_batch) for the CRF and the stand-alone layer entry points below for the rest — which is
 * what the Python mirror's supervision_step does for them (dsrg_amd/ops.py). */
#define DSRG_CTX_MAX_PIXELS 4488
int dsrg_ctx_create(int max_batch, int C, int H, int W, dsrg_ctx_t *out);
int dsrg_ctx_destroy(dsrg_ctx_t ctx);

/* CRFLayer.forward (pylayers/pylayers/pylayers.py:63-88) and, identically,
 * DSRGLayer.refinement (pylayers.py:310-331):
 *   probs_dev   (B,C,H,W) f32   clipped IN PLACE to >= 1e-4 (pylayers.py:67,312)
 *   images_dev  (B,3,Hi,Wi) f32 mean-subtracted BGR (bottom[1]); resampled to
 *               (H,W) with the align-corners order-1 zoom, + mean pixel
 *               (104,117,123), round-half-even, cast to uint8 (pylayers.py:70-75)
 *   refined_dev (B,C,H,W) f64   clip(Q,1e-4)/sum  (pylayers.py:84-86) = self.result
 *   logq_dev    (B,C,H,W) f32   log(refined)      (pylayers.py:88); may be NULL
 * images_dev may be NULL when dsrg_crf_prepare_batch has been called for this batch (see below). */
int dsrg_crf_refine_batch(dsrg_ctx_t ctx, int B, float *probs_dev, const float *images_dev,
                          int img_h, int img_w, const dsrg_crf_params *params,
                          double *refined_dev, float *logq_dev, void *stream);
/* The image-dependent half of the CRF — resampling the images to (H,W) and building the bilateral
 * lattices (Permutohedral::init, CRF/src/permutohedral.cpp:140-321) — needs no network output, so a
 * trainer can run it on a side stream underneath the backbone forward.  After this call (and a
 * stream dependency set up by the caller) dsrg_supervision_step may be given images_dev = NULL.
 * The call overwrites the context's lattices: the caller also orders it BEHIND the last call that
 * reads them (the previous step's mean field), e.g. side stream waits on the main stream first.
 * One host synchronisation of `stream` happens on the FIRST call per (shape, kernel widths): the spatial lattice is built
 * once and its "pixel-local" flag is read back so that later filter launches leave its workgroups out of the grid; inside
 * a stream capture the read-back is postponed to the first later call outside one. */
int dsrg_crf_prepare_batch(dsrg_ctx_t ctx, int B, const float *images_dev, int img_h, int img_w,
                           const dsrg_crf_params *params, void *stream);

/* the same mean-field run on caller-prepared inputs (used by krahenbuhl2013.CRF
 * and by tests): unary_dev (B,C,H,W) f32 holds the NEGATED energies -U (i.e.
 * the `unary` argument of CRF.py:28), im_u8_dev (B,H*W,3) uint8, q_dev
 * (B,C,H,W) f32 receives the marginals. */
int dsrg_crf_meanfield_batch(dsrg_ctx_t ctx, int B, const float *neg_unary_dev,
                             const unsigned char *im_u8_dev, const dsrg_crf_params *params,
                             float *q_dev, void *stream);
/* vertex counts of the lattices built by the last refine/meanfield call:
 * m_gauss, and m_bilateral[b] for b < B (host arrays; synchronises the stream) */
int dsrg_ctx_lattice_sizes(dsrg_ctx_t ctx, int B, int32_t *m_gauss_host, int32_t *m_bilateral_host,
                           void *stream);

/* measurement: per lattice, the number of splat entries beyond the first entry of their vertex (the part of the splat that
 * goes through LDS products; bench.py's LDS traffic model).  Host arrays as for dsrg_ctx_lattice_sizes; synchronises. */
int dsrg_ctx_lattice_extras(dsrg_ctx_t ctx, int B, int32_t *x_gauss_host, int32_t *x_bilateral_host, void *stream);
/* measurement: the geometry of one mean-field filter launch over B images as the launcher plans it NOW (label planes per
 * bilateral / per Gaussian workgroup, workgroups in the grid, dynamic LDS bytes) — bench.py's LDS model reads it instead of
 * re-deriving the launcher's rules.  Host pointers, any may be NULL. */
int dsrg_ctx_filter_plan(dsrg_ctx_t ctx, int B, int32_t *planes_bilateral, int32_t *planes_gaussian, int32_t *workgroups,
                         int32_t *lds_bytes);

/* introspection for the parity tests: one lattice as the reference holds it.  kind 0 = the Gaussian lattice (b = 0),
 * kind 1 = the bilateral lattice of image b of the last refine / prepare / meanfield / supervision call.
 *   keys_host [M*d] int16   vertex keys in id order          (HashTable::getKeys, CRF/src/permutohedral.cpp:296-297)
 *   vid_host  [N*(d+1)]     offset_: vertex id of corner r of pixel i, pixel-major   (permutohedral.cpp:272)
 *   bary_host [N*(d+1)]     barycentric_                                             (permutohedral.cpp:274)
 *   n1_host / n2_host [(d+1)*M]  blur_neighbors_[j*M+i].n1 / .n2, -1 = none          (permutohedral.cpp:315-316)
 * Any array may be NULL; *m_host receives M (size the arrays with dsrg_ctx_lattice_sizes first).  Synchronises. */
int dsrg_ctx_lattice_dump(dsrg_ctx_t ctx, int kind, int b, int32_t *m_host, int16_t *keys_host, int32_t *vid_host,
                          float *bary_host, int32_t *n1_host, int32_t *n2_host, void *stream);
/* introspection for the parity tests: norm_host [N] = 1/sqrt(K 1 + 1e-20) of lattice (kind, b) as dsrg_ctx_lattice_dump
 * addresses it (DenseKernel::initLattice, CRF/src/pairwise.cpp:44,54-57).  Synchronises. */
int dsrg_ctx_lattice_norm(dsrg_ctx_t ctx, int kind, int b, float *norm_host, void *stream);
/* introspection for the parity tests: ONE application of the normalised kernel `kind` (0 Gaussian, 1 bilateral),
 * out = norm . K (norm . q)  (DenseKernel::filter, CRF/src/pairwise.cpp:63-80; splat / blur / slice of
 * CRF/src/permutohedral.cpp:529-589), through the kernels and lattices of the inference loop.  q_dev, out_dev (B,C,H,W) f32. */
int dsrg_ctx_filter_once(dsrg_ctx_t ctx, int kind, int B, const float *q_dev, float *out_dev, void *stream);
/* introspection for the parity tests: the float64 marginals (`self.result`, pylayers.py:84-86) the last
 * dsrg_supervision_step thresholded, copied to refined_dev (B,C,H,W) f64 in stream order. */
int dsrg_ctx_read_refined(dsrg_ctx_t ctx, int B, double *refined_dev, void *stream);

/* measurement hook (no reference counterpart): while profiling is on, every launch of the
 * mean-field filter kernel (splat/blur/slice, the dominant kernel) is bracketed by HIP events
 * on the launch stream; _stop synchronises and returns the summed kernel time and launch count. */
int dsrg_ctx_profile_start(dsrg_ctx_t ctx, int max_launches);
int dsrg_ctx_profile_stop(dsrg_ctx_t ctx, double *total_ms_host, int32_t *launches_host);

/* CRFLayer.backward (pylayers.py:90-92): bottom_diff = (1 - refined) * top_diff */
int dsrg_crf_layer_backward(size_t n, const double *refined_dev, const float *top_diff_dev,
                            float *bottom_diff_dev, void *stream);

/* DSRGLayer.forward -> generate_seed -> generate_seed_step (pylayers.py:237-275,
 * 297-304,333-344) incl. CC_lab (CC_labeling_8.py:112-197), given the refined
 * marginals:
 *   labels_dev (B,1,1,C) f32 0/1, cues_dev (B,C,H,W) f32 0/1,
 *   refined_dev (B,C,H,W) f64, seeds_dev (B,C,H,W) f32 0/1 (output),
 *   scratch_dev: B*H*W uint16 of device scratch (the per-pixel classification handed from the
 *   pixel-parallel pass to the per-image growth pass).  C <= 96. */
int dsrg_srg_grow_batch(int B, int C, int H, int W, const float *labels_dev, const float *cues_dev,
                        const double *refined_dev, double th1, double th2, float *seeds_dev,
                        void *scratch_dev, void *stream);

/* SoftmaxLayer.forward / backward (pylayers.py:30-51) */
int dsrg_softmax_forward(int B, int C, int HW, const float *x_dev, float *p_dev, void *stream);
int dsrg_softmax_backward(int B, int C, int HW, const float *x_dev, const float *top_diff_dev,
                          float *bottom_diff_dev, void *stream);

/* BalancedSeedLossLayer.forward / backward (pylayers.py:126-152).
 * loss_dev: one float.  grad_dev may be NULL (forward only). */
int dsrg_seed_loss(int B, int C, int HW, const float *probs_dev, const float *seeds_dev,
                   float *loss_dev, float *grad_dev, void *stream);
/* ConstrainLossLayer.forward / backward (pylayers.py:160-180); grads may be NULL. */
int dsrg_constrain_loss(int B, int C, int HW, const float *probs_dev, const float *logq_dev,
                        float *loss_dev, float *grad_probs_dev, float *grad_logq_dev, void *stream);

/* ---- the pylayers classes no seed_mc prototxt references, and the evaluation histogram ---------------- */
/* SeedLossLayer (pylayers/pylayers/pylayers.py:94-118): L = -mean_b[ sum S log p / sum S ]; loss and/or grad may be NULL. */
int dsrg_seed_loss_plain(int B, int C, int HW, const float *probs_dev, const float *seeds_dev, float *loss_dev,
                         float *grad_dev, void *stream);
/* ExpandLossLayer (pylayers.py:183-233): probs (B,C,HW), class 0 = background; stat (B,C) image-level labels (> 0.5 =
 * present; stat[:,0] ignored); q_fg / q_bg = 0.996 / 0.999 in the reference.  HW <= 8192.  scratch_dev: B*C doubles
 * (needed when loss_dev != NULL).  loss and/or grad may be NULL. */
int dsrg_expand_loss(int B, int C, int HW, const float *probs_dev, const float *stat_dev, double q_fg, double q_bg,
                     float *loss_dev, float *grad_dev, void *scratch_dev, void *stream);
/* ConfusionMatrix.add (rule_lt = 0: pixels with gt != 255) / generateM (rule_lt = 1: gt < nclass) of
 * training/tools/evaluate.py:25-30,61-68.  hist_dev: nclass*nclass + 1 unsigned 64-bit counters that are ADDED to
 * (row = ground truth, column = prediction); the last one counts kept pixels whose gt or prediction is >= nclass.
 * nclass <= 127. */
int dsrg_confusion_matrix(size_t n, const unsigned char *gt_dev, const unsigned char *pred_dev, int nclass, int rule_lt,
                          unsigned long long *hist_dev, void *stream);

/* Backbone plumbing (no reference counterpart; Caffe's im2col lives in the external framework): NHWC im2col
 * of a 3x3, stride-1, "same"-padded, dilated convolution for 2-byte elements (bf16/fp16), C % 8 == 0:
 *   out[(b,y,x)][tap][c] = in[b][y+(tap/3-1)*dil][x+(tap%3-1)*dil][c], zero outside the map. */
int dsrg_im2col3x3_nhwc16(const void *in_dev, void *out_dev, int B, int H, int W, int C, int dilation, void *stream);
/* Adjoint of dsrg_im2col3x3_nhwc16 for bf16: out[b,y,x,:] = sum over the 9 taps of cols[(b, y-dy*dil, x-dx*dil), tap, :]
 * (cols: (B*H*W, 9*C) row-major, f32 accumulation).  With cols = g @ W^T this is the data gradient of the convolution. */
int dsrg_col2im3x3_nhwc_bf16(const void *cols_dev, void *out_dev, int B, int H, int W, int C, int dilation, void *stream);
/* ReLU backward fused with the bias-gradient reduction of the convolution in front of it: g, y (the ReLU output) and
 * gm are (rows, C) bf16 row-major (NHWC activations), C % 8 == 0; gm = scale * g where y > 0 else 0; bias_grad[c] =
 * sum_r gm[r,c] (f32, summed in a fixed order).  scale = 1 for a plain ReLU; with y = dropout(relu(.)) and
 * scale = 1/(1-p) the same pass is the backward of ReLU + Dropout (y > 0 is both masks at once).  y_dev = gm_dev = NULL:
 * no ReLU, only the column sums of g (any C % 8 == 0 up to 2048; dsrg_bias_grad_bf16 covers the other widths).
 * partials: device scratch of partial_blocks * C floats. */
int dsrg_relu_bwd_bias_bf16(const void *g_dev, const void *y_dev, void *gm_dev, float *bias_grad_dev, float *partials_dev,
                            int partial_blocks, long rows, int C, float scale, void *stream);
/* Column sums of a (rows, C) bf16 matrix in f32, any C <= 256 (bias gradient of a convolution without a ReLU behind it,
 * e.g. the 21-channel fc8 outputs).  partials: device scratch of partial_blocks * C floats. */
int dsrg_bias_grad_bf16(const void *g_dev, float *bias_grad_dev, float *partials_dev, int partial_blocks, long rows, int C,
                        void *stream);
/* Layer-level prototype (round 6) of a FLOAT32 convolution on the bf16 MFMA — the arithmetic of the reference's Caffe
 * convolutions (train-s.prototxt:41-744) at six bf16 products per multiply-add: x = x0 + x1 + x2, w = w0 + w1 + w2 (each term the
 * bf16 rounding of what the terms before leave), products x0 w0, x0 w1, x1 w0, x0 w2, x2 w0, x1 w1 on the fp32 accumulators.
 *   x3_dev (B,H,W,3*cin) bf16: channel plane * cin + c;  w_dev (cout, 6*cin/64, k*k, 64) bf16: the dsrg_conv_igemm_bf16 packing of
 *   the six-block virtual kernel [w0 | w1 | w0 | w2 | w0 | w1];  y_dev (B,H,W,cout) float32.  Forward only, one group. */
int dsrg_conv_igemm_split_f32(const void *x3_dev, const void *w_dev, const float *bias_dev, float *y_dev, int dilation, int B, int H,
                              int W, int cin, int cout, int ksize, int relu, void *stream);
/* The tail of a ResNet bottleneck (the train-f stage on the DeepLab-v2 ResNet-101 of BASELINE.json configs[4]; the reference
 * trains VGG16 only, train-f.prototxt:15-720 — backbone plumbing): y = relu(a + b) over n bf16 elements (fp32 sum, one
 * rounding), n % 8 == 0; and its backward gm = (y > 0) ? g (+ g2) : 0 — g2_dev (may be NULL): a second gradient of y to be
 * added first (the sum autograd would otherwise take in a pass of its own). */
int dsrg_add_relu_bf16(const void *a_dev, const void *b_dev, void *y_dev, size_t n, void *stream);
int dsrg_relu_mask_bf16(const void *g_dev, const void *g2_dev, const void *y_dev, void *gm_dev, size_t n, void *stream);
/* 3x3 / stride 1 / pad 1 average pooling over padded windows (Caffe AVE pooling, pool5a of train-s.prototxt), NHWC bf16,
 * C % 8 == 0.  The stencil is symmetric: the backward pass is the same call on the output gradient. */
int dsrg_avgpool3x3_s1_bf16(const void *in_dev, void *out_dev, int B, int H, int W, int C, void *stream);
/* Direct 3x3 / stride 1 / pad 1 convolution for the narrow layers at the large resolutions: cin, cout in {64, 128}
 * (conv1_2 at 321x321, conv2_1 / conv2_2 at 161x161 of train-s.prototxt:65-160) and 3 -> 64 (conv1_1, :44-64), NHWC bf16 in and out, fp32
 * accumulation, optional bias (cout f32) and ReLU in the epilogue:
 *   y[b,y,x,o] = relu?( bias[o] + sum_{dy,dx,c} w[o][dy+1][dx+1][c] * x[b,y+dy,x+dx,c] )
 * w_dev: (cout, 3, 3, cin) bf16 = the memory of a channels_last (out, in, 3, 3) tensor.  With the kernel flipped and its
 * channel axes swapped the same call is the data gradient of that convolution.  Other channel counts: DSRG_ERR_INVALID. */
int dsrg_conv3x3_direct_bf16(const void *x_dev, const void *w_dev, const float *bias_dev, void *y_dev, int B, int H, int W,
                             int cin, int cout, int relu, void *stream);
/* The data-gradient form with the ReLU backward of the layer below in its store (cin, cout in {64, 128}): gx = conv(g, w) where
 * mask_dev (B,H,W,cout) bf16 — that layer's output — is positive, else 0; bias_grad_dev (cout f32) = sum over pixels of gx
 * (fixed summation order).  workspace_dev: dsrg_conv3x3_direct_dgrad_workspace(cout) bytes. */
size_t dsrg_conv3x3_direct_dgrad_workspace(int cout);
int dsrg_conv3x3_direct_dgrad_bf16(const void *g_dev, const void *w_dev, const void *mask_dev, void *gx_dev, float *bias_grad_dev,
                                   void *workspace_dev, size_t workspace_bytes, int B, int H, int W, int cin, int cout, void *stream);
/* Weight gradient of the same convolution for (cin, cout) in {(3, 64), (64, 64), (64, 128), (128, 128)}:
 *   gw[o][dy+1][dx+1][c] = sum_{b,y,x} g[b,y,x,o] * x[b,y+dy,x+dx,c]      (zero padding)
 * x_dev (B,H,W,cin) and g_dev (B,H,W,cout) NHWC bf16; gw_dev (cout, 3, 3, cin) bf16 = the memory of a channels_last
 * (out, in, 3, 3) tensor; fp32 accumulation, summed in a fixed order (deterministic).  workspace_dev: device scratch of
 * dsrg_conv3x3_wgrad_workspace(B, H, W, cin, cout) bytes (0 = unsupported channel counts). */
size_t dsrg_conv3x3_wgrad_workspace(int B, int H, int W, int cin, int cout);
int dsrg_conv3x3_wgrad_bf16(const void *x_dev, const void *g_dev, void *gw_dev, void *workspace_dev, size_t workspace_bytes,
                            int B, int H, int W, int cin, int cout, void *stream);
/* The same with the gradient in float32 (the master parameters' dtype: no rounding, no cast pass). */
int dsrg_conv3x3_wgrad_f32(const void *x_dev, const void *g_dev, float *gw_dev, void *workspace_dev, size_t workspace_bytes, int B,
                           int H, int W, int cin, int cout, void *stream);
/* Both bf16 kernels dsrg_conv3x3_direct_bf16 reads, from the float32 master (cout, 3, 3, cin) in one pass (cin, cout in {64, 128}):
 * fwd_dev (cout, 3, 3, cin) — the parameter cast — and dgrad_dev (cin, 3, 3, cout) — flipped, channel axes swapped: the kernel of
 * the data gradient.  Either may be NULL. */
int dsrg_pack_conv_weight_direct_f32(const float *w_dev, void *fwd_dev, void *dgrad_dev, int cout, int cin, void *stream);
/* Implicit-GEMM convolution for the wide layers (conv3_x, conv4_x, conv5_x, fc6_k, fc7_k of train-s.prototxt:161-736): 3x3
 * with any dilation ('same' zero padding) or 1x1, stride 1, cin % 64 == 0, cout % 256 == 0, NHWC bf16 in and out, fp32
 * accumulation, optional bias (cout f32) and ReLU in the epilogue (the launch also takes cout = 64 — half-empty channel tiles, for
 * bandwidth-bound 1x1 layers; dsrg_conv_igemm_supported names the shapes it is the recommended route for); no im2col matrix is formed:
 *   y[b,y,x,o] = relu?( bias[o] + sum_{dy,dx,c} w[o][dy+1][dx+1][c] * x[b,y+dy*dil,x+dx*dil,c] )
 * Up to four independent problems of one geometry (the four ASPP branches) share a launch: x_dev, w_dev, bias_dev (may be
 * NULL, entries may be NULL), y_dev and dilation are HOST arrays of ngroups entries.  w_dev[g]: the kernel packed as
 * (cout, cin / 64, ksize * ksize, 64) bf16 — w_packed[o][cc][tap][c] = w[o][cc * 64 + c][tap / 3][tap % 3].  With the kernel
 * flipped and its channel axes swapped the same call is the data gradient.  dropout_p > 0: the Dropout layer behind the
 * ReLU (train-s.prototxt: drop6_k, drop7_k) in the same epilogue — y = relu(..) * keep / (1 - p), keep a pure function of
 * (dropout_seed, group, element position) from a counter-based generator, p realised in steps of 1/256; the backward pass
 * reads both masks off the sign of y.  dsrg_conv_igemm_supported: 1 if the channel counts / kernel size are served, else 0
 * (the call then returns DSRG_ERR_UNSUPPORTED).
 * workspace_dev (may be NULL): dsrg_conv_igemm_workspace() bytes of device scratch owned by the caller and used by ONE stream
 * at a time.  With it, a launch whose tiles do not fill whole rounds of the chip deals its K-steps out evenly instead
 * ("stream-K": one workgroup per CU, partial tiles handed over through the scratch, summed in a fixed order — results are
 * deterministic, equal to the whole-tile launch up to the fp32 summation order of a cut tile).
 * dsrg_conv_igemm_workspace_status (tests): 0, or 1 if a workgroup of the last launch on that scratch gave up waiting. */
int dsrg_conv_igemm_supported(int cin, int cout, int ksize);
int dsrg_conv_igemm_bf16(const void *const *x_dev, const void *const *w_dev, const float *const *bias_dev, void *const *y_dev,
                         const int *dilation, int ngroups, int B, int H, int W, int cin, int cout, int ksize, int relu,
                         float dropout_p, unsigned long long dropout_seed, void *workspace_dev, size_t workspace_bytes,
                         void *stream);
size_t dsrg_conv_igemm_workspace(void);
/* The same launch as the data gradient of a convolution whose INPUT was the output y_below of a ReLU (optionally followed by
 * Dropout) layer: gx = conv(g, w_dgrad) * mask_scale where mask = y_below > 0, else 0 — the ReLU / Dropout backward folded
 * into the store — and bias_grad[g][c] = sum over pixels of gx (the bias gradient of the layer below; fp32, fixed summation
 * order).  g: (B, H, W, cin) gradient of this layer's output; w: the flipped + transposed packing of dsrg_pack_conv_weight_f32
 * (cout = the layer's INPUT channels); mask, gx: (B, H, W, cout) bf16; bias_grad_dev: NULL or ngroups pointers to cout floats;
 * workspace_dev: dsrg_conv_igemm_dgrad_workspace(...) bytes when bias_grad_dev is given.  Replaces the separate
 * relu-backward + bias-sum pass over the gradient (mirror: torch.autograd of Conv -> ReLU -> Dropout in one step). */
size_t dsrg_conv_igemm_dgrad_workspace(int ngroups, int B, int H, int W, int cout);
int dsrg_conv_igemm_dgrad_bf16(const void *const *g_dev, const void *const *w_dev, const void *const *mask_dev, void *const *gx_dev,
                               float *const *bias_grad_dev, const int *dilation, int ngroups, int B, int H, int W, int cin,
                               int cout, int ksize, float mask_scale, void *workspace_dev, size_t workspace_bytes, void *stream);
/* One convolution of the same family with a RESIDUAL in its store (a ResNet bottleneck's shortcut; the reference's stage-2 networks
 * are DeepLab-v2 VGG16 / ResNet-101 — training/experiment/anti-noise/config/deeplabv2_weak.prototxt — where Caffe runs Eltwise SUM +
 * ReLU layers of their own):  y = post( bf16(conv(x, w) + bias) + res ),  post = ReLU (relu != 0) and / or zero where mask <= 0
 * (mask_dev may be NULL).  res, mask, y: (B, H, W, cout) bf16.  Bit for bit what dsrg_conv_igemm_bf16 followed by dsrg_add_relu_bf16
 * (forward: res = the shortcut) or by a bf16 add and dsrg_relu_mask_bf16 (data gradient of the block's first convolution: res = the
 * gradient that reaches the block input along the shortcut, mask = the block input, itself a ReLU output) gives, in one launch. */
int dsrg_conv_igemm_residual_bf16(const void *x_dev, const void *w_dev, const float *bias_dev, const void *res_dev, const void *mask_dev,
                                  void *y_dev, int dilation, int B, int H, int W, int cin, int cout, int ksize, int relu, void *stream);
/* The DeepLab-v2 ASPP head — J = branches x 9 (branch, tap) pairs of dilated 3x3 classifiers with `outputs` (21) channels each, all
 * reading ONE feature map and summed (deeplabv2 prototxt: fc1_voc12_c0..c3 + Eltwise SUM) — as one 1x1 convolution plus a shifted
 * gather: y = dsrg_conv_igemm_bf16(x, the (J outputs, rounded up to `channels`) x cin matrix of all kernels' taps, 1x1) and
 *   out[b,y,x,o] = bias[o] + sum_j y[b, y + dy_j, x + dx_j, j outputs + o]        (zero outside the map; j ascending; fp32)
 * (dsrg_aspp_shift_sum_f32; y (B,H,W,channels) bf16, out (B,H,W,outputs) f32, bias_dev (outputs) or NULL = the branches' biases
 * summed).  Backward: gp[b,y,x,j outputs + o] = bf16(g[b, y - dy_j, x - dx_j, o]) (dsrg_aspp_shift_gather_bf16; channels past
 * J outputs zero) is the gradient of y; the 1x1 layer's data / weight gradients follow.  offsets: HOST array of npairs (dy, dx)
 * pairs, npairs <= 36.  No output channel is padded to a 128-wide tile and the feature map is read once per pass. */
int dsrg_aspp_shift_sum_f32(const void *y_dev, const float *bias_dev, float *out_dev, const int *offsets, int npairs, int outputs,
                            int channels, int B, int H, int W, void *stream);
int dsrg_aspp_shift_gather_bf16(const float *g_dev, void *gp_dev, const int *offsets, int npairs, int outputs, int channels, int B, int H,
                                int W, void *stream);
/* Deferred bias-gradient reductions.  The launches that hand back a bias gradient (dsrg_conv_igemm_dgrad_bf16 / dsrg_conv_igemm_backward_bf16
 * with bias_grad_dev, dsrg_conv3x3_direct_dgrad_bf16, dsrg_heads_backward_relu_bf16, dsrg_relu_bwd_bias_bf16, dsrg_bias_grad_bf16,
 * dsrg_maxpool3x3_bwd_relu_bf16) end with a pass of 5-7 us that sums per-tile / per-block partial rows (fifteen such passes per
 * train-s step, 96 us) and nothing reads a bias gradient before the update.  dsrg_defer_reductions(1): from now on these passes
 * are RECORDED instead of launched (up to 48; beyond that they run as before); dsrg_flush_reductions(stream) runs everything
 * recorded in ONE launch on `stream` — the same arithmetic in the same order, the gradients' bits do not change — and
 * dsrg_defer_reductions(0) stops recording.  Process-wide.  The caller's duties while recording: the partial-row scratch it passed
 * to each launch (workspace_dev / partial_dev) stays alive, and is not shared between launches, until the flush has been enqueued
 * on the stream the launches ran on; nothing reads the bias gradients before it. */
int dsrg_defer_reductions(int on);
int dsrg_flush_reductions(void *stream);
int dsrg_conv_igemm_workspace_status(const void *workspace_dev, void *stream, int *status_host);
/* The two packed forms dsrg_conv_igemm_bf16 reads, from the float32 master kernel in ONE pass (cast included): w_dev
 * (cout, ksize*ksize, cin) f32 = the memory of a channels_last (cout, cin, ksize, ksize) parameter; fwd_dev (may be NULL):
 * (cout, cin / 64, taps, 64) bf16 for the forward; dgrad_dev (may be NULL): (cin, cout / 64, taps, 64) bf16 for the data
 * gradient (kernel flipped, channel axes swapped).  64 | cout, 64 | cin. */
int dsrg_pack_conv_weight_f32(const float *w_dev, void *fwd_dev, void *dgrad_dev, int cout, int cin, int ksize, void *stream);
/* The same with a constant per-output-channel factor folded in: w[o] * scale_dev[o] is what is packed (scale_dev (cout) f32, may be
 * NULL = dsrg_pack_conv_weight_f32) — a frozen BatchNorm's scale behind the convolution (DeepLab-v2 ResNet-101, use_global_stats). */
int dsrg_pack_conv_weight_scaled_f32(const float *w_dev, const float *scale_dev, void *fwd_dev, void *dgrad_dev, int cout, int cin,
                                     int ksize, void *stream);
/* Caffe's SGDSolver update (solver-s.prototxt:5-14: momentum 0.9, weight_decay 5e-4, per-blob lr_mult / decay_mult of
 * train-s.prototxt) of n float32 parameters, sixteen per launch, in the form  B <- momentum B + (g + wd W);  W <- W - lr B
 * (B = Caffe's history / lr), WITH the packed bf16 kernels of the convolution routes written from the new values in the same
 * pass (the packs of dsrg_pack_conv_weight_f32 / dsrg_pack_conv_weight_direct_f32 — the next forward then reads no float32
 * weight).  Host arrays of n entries each: param_dev / grad_dev / momentum_dev device pointers (4-byte aligned; parameter and momentum of a packed kernel 16, same element
 * order; grad_dev[i] = NULL: tensor i is only packed); fwd_dev[i] / dgrad_dev[i] packed outputs or NULL (the arrays themselves
 * may be NULL: nothing packed); shape[4 i ..] = {cout, cin, taps (1 | 9), plain (1: the direct kernels' layouts)} of a packed
 * tensor, whose memory is (cout, taps, cin), 64 | cout, 64 | cin; numel[i]; lr[i] = base_lr * lr_mult, weight_decay[i] =
 * weight_decay * decay_mult.  Element-wise, one thread owns an element: deterministic. */
int dsrg_sgd_pack_f32(int n, float *const *param_dev, const float *const *grad_dev, float *const *momentum_dev, void *const *fwd_dev,
                      void *const *dgrad_dev, const int *shape, const long long *numel, const float *lr, const float *weight_decay,
                      float momentum, void *stream);
/* Weight gradient of the same convolutions, again without an im2col matrix (cin % 64 == 0, cout % 64 == 0, ksize 1 or 3; tiles of 256
 * outputs x 256 inputs of one tap — full, hence the recommended route, where cin % 256 == 0 (or cin = 128 with a 3x3 kernel) and
 * cout % 256 == 0; narrower tensors run partly empty tiles, which is fine where the layer is bandwidth-bound):
 *   gw[o][tap][c] = sum_{b,y,x} g[b,y,x,o] * x[b,y+dy*dil,x+dx*dil,c]      (zero padding; tap = 3 (dy+1) + dx+1)
 * x_dev[g] (B,H,W,cin) and g_dev[g] (B,H,W,cout) NHWC bf16; gw_dev[g] (cout, ksize*ksize, cin) = the memory of a channels_last
 * (cout, cin, ksize, ksize) tensor, float32 (out_bf16 = 0: the master weights' gradient, no cast behind it) or bf16; fp32
 * accumulation; the pixel range is split over workgroups whose partial tiles are summed in a fixed order (deterministic).
 * x_dev, g_dev, gw_dev, dilation: HOST arrays of ngroups (1..4) entries (the four ASPP branches in one launch).
 * workspace_dev: dsrg_conv_igemm_wgrad_workspace(ngroups, B, H, W, cin, cout, ksize) bytes of device scratch (0 = unsupported). */
size_t dsrg_conv_igemm_wgrad_workspace(int ngroups, int B, int H, int W, int cin, int cout, int ksize);
int dsrg_conv_igemm_wgrad_bf16(const void *const *x_dev, const void *const *g_dev, void *const *gw_dev, const int *dilation,
                               int ngroups, void *workspace_dev, size_t workspace_bytes, int B, int H, int W, int cin, int cout,
                               int ksize, int out_bf16, void *stream);
/* The whole backward of ONE 3x3 convolution of the forward geometry cin -> cout as one launch: the data gradient gx (B,H,W,cin) of g
 * (B,H,W,cout) with the data-gradient packing w_dgrad_dev (dsrg_pack_conv_weight_f32) — with mask_dev (the layer's INPUT = the ReLU
 * output of the layer below, (B,H,W,cin) bf16) and bias_grad_dev given also that layer's ReLU (+ Dropout: mask_scale) backward and
 * bias gradient, as dsrg_conv_igemm_dgrad_bf16 — and the weight gradient gw (cout, cin, k, k; float32, channels_last) from x_dev
 * (B,H,W,cin) and g.  The data gradient's tiles and the weight gradient's workgroups share a grid, so the CUs a 212-tile data
 * gradient leaves idle take weight-gradient work; where that form does not apply (dilation >= 3, a 128-channel x) the two launches
 * run one after the other.  The data gradient equals dsrg_conv_igemm_dgrad_bf16 / dsrg_conv_igemm_bf16 bit for bit; the weight gradient
 * equals dsrg_conv_igemm_wgrad_bf16 up to fp32 reassociation (its pixel split is chosen for the merged grid); both deterministic.
 * Workspaces: dsrg_conv_igemm_dgrad_workspace(1, B, H, W, cin) bytes (only with bias_grad_dev) and
 * dsrg_conv_igemm_wgrad_workspace(1, B, H, W, cin, cout, ksize) bytes. */
int dsrg_conv_igemm_backward_bf16(const void *g_dev, const void *w_dgrad_dev, const void *x_dev, const void *mask_dev, void *gx_dev,
                                  float *gw_dev, int dilation, float *bias_grad_dev, float mask_scale, void *colsum_workspace_dev,
                                  size_t colsum_workspace_bytes, void *wgrad_workspace_dev, size_t wgrad_workspace_bytes, int B, int H,
                                  int W, int cin, int cout, int ksize, void *stream);
/* The same whole backward (3x3 with dilation < 3, or 1x1: merged grid; else two launches) for a convolution inside a residual block
 * with a folded constant scale: res_dev (may be NULL; (B,H,W,cin) bf16) is added to the bf16-rounded data gradient before the mask, as
 * dsrg_conv_igemm_residual_bf16; gw_scale_dev (may be NULL; (cout) f32): gw[o] = scale[o] * (the gradient of the scaled kernel the
 * forward ran with).  No bias gradient.  mask_dev may be NULL. */
int dsrg_conv_igemm_backward_residual_bf16(const void *g_dev, const void *w_dgrad_dev, const void *x_dev, const void *mask_dev,
                                           const void *res_dev, void *gx_dev, float *gw_dev, const float *gw_scale_dev, int dilation,
                                           void *wgrad_workspace_dev, size_t wgrad_workspace_bytes, int B, int H, int W, int cin, int cout,
                                           int ksize, void *stream);
/* The four fc8-SEC_k 1x1 classifiers and their Eltwise SUM (train-s.prototxt:461-744) in one pass with float32 weights,
 * float32 accumulation and a float32 NCHW result: out[b][o][hw] = sum_k ( x_k[(b,hw)][:] . w[k][o][:] + bias[k][o] ).
 * x_dev: host array of n_branches (<= 4) device pointers to (B*HW, K) bf16 row-major (NHWC) activations; w_dev
 * (n_branches, O, K) f32; bias_dev (n_branches, O) f32 or NULL; out_dev (B, O, HW) f32.  O <= 32, K % 256 == 0. */
int dsrg_heads_forward_bf16(const void *const *x_dev, int n_branches, const float *w_dev, const float *bias_dev,
                            float *out_dev, int B, int HW, int K, int O, void *stream);
/* Its backward from the float32 score gradient g_dev (B, O, HW):
 *   gx_dev (may be NULL): n_branches matrices (B*HW, K) bf16 row-major, branch k at byte offset k * gx_branch_stride_bytes:
 *                         gx_k[m][c] = sum_o g[m][o] w[k][o][c]                       (float32 accumulation)
 *   gw_dev (may be NULL): (n_branches, O, K) f32: gw[k][o][c] = sum_m g[m][o] x_k[m][c] (exact f32 fma chains, fixed order);
 *                         partial_dev: scratch of dsrg_heads_backward_chunks(B*HW) * n_branches * O * K floats. */
int dsrg_heads_backward_chunks(int M);
int dsrg_heads_backward_bf16(const void *const *x_dev, int n_branches, const float *w_dev, const float *g_dev,
                             void *gx_dev, size_t gx_branch_stride_bytes, float *gw_dev, float *partial_dev, int B,
                             int HW, int K, int O, void *stream);
/* The same with the backward of the ReLU (+ Dropout) layers that produced the x_k folded into the data gradient's store:
 * gx_k[m][c] = relu_scale * sum_o g[m][o] w[k][o][c] where x_k[m][c] > 0, else 0, and bias_grad_dev (n_branches, K) f32 =
 * sum over m of gx_k (the bias gradient of the layer below; fixed summation order).  workspace_dev:
 * dsrg_heads_backward_relu_workspace(n_branches, B*HW, K) bytes.  gw as above (from the unmasked x_k, g). */
size_t dsrg_heads_backward_relu_workspace(int n_branches, int M, int K);
int dsrg_heads_backward_relu_bf16(const void *const *x_dev, int n_branches, const float *w_dev, const float *g_dev,
                                  void *gx_dev, size_t gx_branch_stride_bytes, float *gw_dev, float *partial_dev,
                                  float relu_scale, float *bias_grad_dev, void *workspace_dev, size_t workspace_bytes, int B,
                                  int HW, int K, int O, void *stream);
/* 3x3 max pooling, pad 1, stride 1 or 2, NHWC bf16 (the Pooling layers of train-s.prototxt:69-80 etc.; OH/OW chosen by
 * the caller, ceil mode included).  code_dev: B*OH*OW*C bytes, the window position (3*dy+dx) of the first maximum. */
int dsrg_maxpool3x3_fwd_bf16(const void *in_dev, void *out_dev, void *code_dev, int B, int H, int W, int OH, int OW, int C,
                             int stride, void *stream);
/* The same pooling of a ReLU's OUTPUT when the pool's backward is to carry that ReLU's backward as well: windows whose maximum
 * is not positive get the code 0xfe, which names no window position — their gradient goes nowhere, which is what masking the
 * pooled-back gradient with (input > 0) does (a pixel receives gradient only as the argmax of a window, and its value is
 * that window's maximum).  Pooled values are the same as dsrg_maxpool3x3_fwd_bf16's. */
int dsrg_maxpool3x3_relu_fwd_bf16(const void *in_dev, void *out_dev, void *code_dev, int B, int H, int W, int OH, int OW, int C,
                                  int stride, void *stream);
int dsrg_maxpool3x3_bwd_bf16(const void *gout_dev, const void *code_dev, void *gin_dev, int B, int H, int W, int OH, int OW,
                             int C, int stride, void *stream);
/* Stride-2 pooling backward fused with the ReLU backward and the bias gradient of the convolution in front of the pool
 * (conv + ReLU + pool, train-s.prototxt:65-226): relu_out_dev = the pool's input = that ReLU's output (B,H,W,C) bf16;
 * gin = (relu_out > 0) ? pooled-back gradient : 0 (bit-identical to dsrg_maxpool3x3_bwd_bf16 followed by
 * dsrg_relu_bwd_bias_bf16), bias_grad_dev[c] = sum of gin over (b, y, x).  partials_dev: partial_blocks * C floats;
 * (C / 8) must divide 256.  relu_out_dev may be NULL when code_dev comes from dsrg_maxpool3x3_relu_fwd_bf16: the codes then
 * carry the mask, the pool's input (211 MB at pool1, batch 16) is not read again, and the result is the same bit for bit. */
int dsrg_maxpool3x3_bwd_relu_bf16(const void *gout_dev, const void *code_dev, const void *relu_out_dev, void *gin_dev,
                                  float *bias_grad_dev, float *partials_dev, int partial_blocks, int B, int H, int W, int OH,
                                  int OW, int C, void *stream);

/* The five Python layers of train-s.prototxt:746-810 as ONE stream-ordered
 * sequence (Softmax -> CRF -> DSRG -> BalancedSeedLoss + ConstrainLoss, then
 * the backward pass of A.3 down to d loss / d fc8), computing the CRF once
 * (SURVEY §0.2).  Outputs:
 *   losses_dev[2]   = {loss-Seed, loss-Constrain}
 *   grad_logits_dev = d(loss-Seed + loss-Constrain)/d fc8-SEC      (B,C,H,W)
 *   probs_dev / seeds_dev / logq_dev: optional (may be NULL) copies of the
 *   clipped softmax blob, the grown seeds and the CRF log-marginals.
 * images_dev may be NULL when dsrg_crf_prepare_batch has been called for this batch. */
int dsrg_supervision_step(dsrg_ctx_t ctx, int B, const float *logits_dev, const float *images_dev,
                          int img_h, int img_w, const float *labels_dev, const float *cues_dev,
                          double th1, double th2, const dsrg_crf_params *params,
                          float *losses_dev, float *grad_logits_dev,
                          float *probs_dev, float *seeds_dev, float *logq_dev, void *stream);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif /* DSRG_HIP_H */
