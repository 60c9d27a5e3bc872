/*
 * oracle/dsrg_oracle.c — CPU restatement of the DSRG supervision hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (dsrg_amd/, pylayers/,
 * krahenbuhl2013/, libdsrg_hip.so) may include, link, import or call this file.
 * It is used by tests/ (as the checker), by __graft_entry__.smoke() (as the
 * checker) and by bench.py's `cpu_baseline` leg (as the timed CPU baseline).
 *
 * PARITY STATUS
 *   - Seeded region growing (orc_srg_grow) and connected components
 *     (orc_cc_label8): PINNED.  Checked bit-exact against the reference's own
 *     Python (`generate_seed_step`, `CC_lab`) run in the build container; the
 *     input/output vectors are committed under tests/golden/ together with the
 *     script that produced them (tests/golden/make_golden.py).
 *   - Dense CRF (lattice / filter / mean field): PARITY UNPINNED.  The reference
 *     C++ (the .cpp files under CRF/src) needs Eigen3, an un-vendored external dependency that
 *     is absent from this image, so it cannot be built here, and the reference
 *     holds no tests or golden vectors.  This file restates the algorithm from
 *     the reference sources line by line (citations below), following the SSE
 *     code path that an x86 build takes; it is checked against analytic
 *     known-answer properties (rows sum to 1, label-permutation equivariance,
 *     identity kernel at sub-pixel sigma, brute-force Gaussian agreement) and
 *     against the reference's SECOND formulation of the lattice — the scalar
 *     Permutohedral::init of permutohedral.cpp:323-474, restated below as
 *     orc_lattice_init_scalar and used by nothing but that cross-check
 *     (tests/test_oracle_golden.py): same vertex keys, same per-pixel weights,
 *     marginals within 5e-6.  Still unpinned: both are restatements.
 *   - Softmax / BalancedSeedLoss / ConstrainLoss: PARITY UNPINNED (Theano is
 *     absent); closed forms derived from the Theano expressions and verified
 *     against torch autograd in fp64 (tests/test_oracle_layers.py).
 *
 * Compile: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 * -ffp-contract=off matters: the reference is built for baseline x86-64 (no
 * FMA), so every a*b+c below is two roundings.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* Permutohedral lattice — follows CRF/src/permutohedral.cpp                  */
/* ------------------------------------------------------------------------- */

typedef struct {
    int N, d, M;
    int *offset;   /* [Npad*(d+1)] vertex id of vertex r of pixel i (permutohedral.cpp:272) */
    float *bary;   /* [Npad*(d+1)] barycentric weight               (permutohedral.cpp:274) */
    int *n1, *n2;  /* [(d+1)*M]    blur neighbours, -1 = none        (permutohedral.cpp:315-316) */
    short *keys;   /* [M*d]        vertex keys in insertion order */
} orc_lattice;

/* HashTable restated (permutohedral.cpp:54-131): open addressing, linear
 * probing, ids handed out in insertion order.  Capacity handling differs
 * (fixed power of two, no grow) — ids and results do not depend on it. */
typedef struct {
    int d, filled, cap;
    short *keys;
    int *table;
} orc_hash;

static size_t orc_hash_fn(const short *k, int d) {
    size_t r = 0;               /* permutohedral.cpp:80-87 */
    for (int i = 0; i < d; i++) { r += (size_t)(long)k[i]; r *= 1664525; }
    return r;
}
static void orc_hash_init(orc_hash *h, int d, int max_elems) {
    int cap = 16;
    while (cap < 4 * max_elems) cap *= 2;
    h->d = d; h->filled = 0; h->cap = cap;
    h->keys = (short *)malloc(sizeof(short) * (size_t)d * (size_t)(max_elems + 1));
    h->table = (int *)malloc(sizeof(int) * (size_t)cap);
    for (int i = 0; i < cap; i++) h->table[i] = -1;
}
static int orc_hash_find(orc_hash *h, const short *k, int create) {
    size_t p = orc_hash_fn(k, h->d) & (size_t)(h->cap - 1);
    for (;;) {
        int e = h->table[p];
        if (e == -1) {
            if (!create) return -1;
            memcpy(h->keys + (size_t)h->filled * h->d, k, sizeof(short) * h->d);
            h->table[p] = h->filled;
            return h->filled++;
        }
        if (memcmp(h->keys + (size_t)e * h->d, k, sizeof(short) * h->d) == 0) return e;
        p = (p + 1) & (size_t)(h->cap - 1);
    }
}

static void orc_lattice_free(orc_lattice *L) {
    free(L->offset); free(L->bary); free(L->n1); free(L->n2); free(L->keys);
    memset(L, 0, sizeof(*L));
}

/* Round to nearest, ties to even — what _mm_cvtps_epi32 / _mm_round_ps do
 * under the default MXCSR (permutohedral.cpp:213-217). */
static float orc_rne(float v) { return nearbyintf(v); }

/* Permutohedral::init, SSE variant (permutohedral.cpp:140-321).
 * feature is d x N column-major: feature[k*d + j] = feature(j,k).
 * The SSE loop walks pixels in blocks of 4 and pads the last block with zero
 * features (permutohedral.cpp:191-196), so when N%4 != 0 the origin simplex is
 * inserted into the hash table even if no pixel lies in it ("phantom"
 * vertices); they receive no splat weight but take part in the blur. */
static void orc_lattice_init(orc_lattice *L, const float *feature, int d, int N) {
    const int Npad = (N + 3) / 4 * 4;
    const int d1 = d + 1;
    L->N = N; L->d = d;
    L->offset = (int *)calloc((size_t)Npad * d1, sizeof(int));
    L->bary = (float *)calloc((size_t)Npad * d1, sizeof(float));

    orc_hash H;
    orc_hash_init(&H, d, Npad * d1);

    short *canonical = (short *)malloc(sizeof(short) * d1 * d1);
    for (int i = 0; i <= d; i++) {                      /* :171-176 */
        for (int j = 0; j <= d - i; j++) canonical[i * d1 + j] = (short)i;
        for (int j = d - i + 1; j <= d; j++) canonical[i * d1 + j] = (short)(i - d1);
    }
    float *scale = (float *)malloc(sizeof(float) * (d > 0 ? d : 1));
    const float inv_std_dev = (float)(sqrt(2.0 / 3.0) * (double)d1);       /* :179 */
    for (int i = 0; i < d; i++)                                            /* :182 */
        scale[i] = (float)(1.0 / sqrt((double)((i + 2) * (i + 1))) * (double)inv_std_dev);

    const float invdplus1 = 1.0f / (float)d1;   /* :148 */
    const float dplus1 = (float)d1;             /* :149 */

    float *f = (float *)malloc(sizeof(float) * (d > 0 ? d : 1));
    float *elevated = (float *)malloc(sizeof(float) * d1);
    float *rem0 = (float *)malloc(sizeof(float) * d1);
    float *rank = (float *)malloc(sizeof(float) * d1);
    float *bc = (float *)malloc(sizeof(float) * (d + 2));
    short *key = (short *)malloc(sizeof(short) * d1);

    for (int k = 0; k < Npad; k++) {
        for (int j = 0; j < d; j++) f[j] = k < N ? feature[(size_t)k * d + j] : 0.0f;   /* :194-196 */

        float sm = 0.0f;                                 /* :201-207 */
        for (int j = d; j > 0; j--) {
            float cf = f[j - 1] * scale[j - 1];
            float jc = (float)j * cf;
            elevated[j] = sm - jc;
            sm = sm + cf;
        }
        elevated[0] = sm;

        float sum = 0.0f;                                /* :210-220 */
        for (int i = 0; i <= d; i++) {
            float v = invdplus1 * elevated[i];
            v = orc_rne(v);
            rem0[i] = v * dplus1;
            sum = sum + v;
        }
        for (int i = 0; i <= d; i++) rank[i] = 0.0f;     /* :223-233 */
        for (int i = 0; i < d; i++) {
            float di = elevated[i] - rem0[i];
            for (int j = i + 1; j <= d; j++) {
                float dj = elevated[j] - rem0[j];
                float c = (di < dj) ? 1.0f : 0.0f;
                rank[i] = rank[i] + c;
                rank[j] = rank[j] + (1.0f - c);
            }
        }
        for (int i = 0; i <= d; i++) {                   /* :236-242 */
            rank[i] = rank[i] + sum;
            float add = (rank[i] < 0.0f) ? dplus1 : 0.0f;
            float sub = (rank[i] >= dplus1) ? dplus1 : 0.0f;
            float as = add - sub;
            rank[i] = rank[i] + as;
            rem0[i] = rem0[i] + as;
        }
        for (int i = 0; i < d + 2; i++) bc[i] = 0.0f;    /* :245-258 */
        for (int i = 0; i <= d; i++) {
            float v = (elevated[i] - rem0[i]) * invdplus1;
            int p = (int)((float)d - rank[i]);
            bc[p] = bc[p] + v;
            bc[p + 1] = bc[p + 1] - v;
        }
        bc[0] = bc[0] + (1.0f + bc[d + 1]);              /* :263 */

        for (int r = 0; r <= d; r++) {                   /* :268-275 */
            for (int i = 0; i < d; i++)
                key[i] = (short)(rem0[i] + (float)canonical[r * d1 + (int)rank[i]]);
            L->offset[(size_t)k * d1 + r] = orc_hash_find(&H, key, 1);
            L->bary[(size_t)k * d1 + r] = bc[r];
        }
    }

    const int M = H.filled;                              /* :296-318 */
    L->M = M;
    L->n1 = (int *)malloc(sizeof(int) * (size_t)d1 * (M > 0 ? M : 1));
    L->n2 = (int *)malloc(sizeof(int) * (size_t)d1 * (M > 0 ? M : 1));
    short *a = (short *)malloc(sizeof(short) * d1);
    short *b = (short *)malloc(sizeof(short) * d1);
    for (int j = 0; j <= d; j++) {
        for (int i = 0; i < M; i++) {
            const short *kk = H.keys + (size_t)i * d;
            for (int k = 0; k < d; k++) { a[k] = (short)(kk[k] - 1); b[k] = (short)(kk[k] + 1); }
            if (j < d) { a[j] = (short)(kk[j] + d); b[j] = (short)(kk[j] - d); }
            /* for j == d the reference writes n1[d]/n2[d], a slot find() never reads */
            L->n1[(size_t)j * M + i] = orc_hash_find(&H, a, 0);
            L->n2[(size_t)j * M + i] = orc_hash_find(&H, b, 0);
        }
    }
    L->keys = H.keys;
    free(H.table);
    free(a); free(b); free(canonical); free(scale); free(f); free(elevated);
    free(rem0); free(rank); free(bc); free(key);
}

/* Permutohedral::init, scalar variant (permutohedral.cpp:323-474) — the reference's second, independent formulation of the
 * same lattice (what a build without SSE runs).  Restated here ONLY as a cross-check of the SSE restatement above
 * (tests/test_oracle_golden.py): the two differ in how they round (ceil/floor comparison, ties go DOWN, instead of
 * round-half-even), in `sum` being an int updated through a float (`:384`), in rank being short arithmetic, in the wrap-around
 * term `1.0 + barycentric[d+1]` being evaluated in double (`:422`), and in the pixel loop not being padded to a multiple of
 * four (no phantom origin vertices).  A transcription slip in either restatement breaks their agreement. */
static void orc_lattice_init_scalar(orc_lattice *L, const float *feature, int d, int N) {
    const int d1 = d + 1;
    L->N = N; L->d = d;
    L->offset = (int *)calloc((size_t)(N > 0 ? N : 1) * d1, sizeof(int));
    L->bary = (float *)calloc((size_t)(N > 0 ? N : 1) * d1, sizeof(float));
    orc_hash H;
    orc_hash_init(&H, d, N * d1 + d1);

    float *scale_factor = (float *)malloc(sizeof(float) * (d > 0 ? d : 1));
    float *elevated = (float *)malloc(sizeof(float) * d1);
    float *rem0 = (float *)malloc(sizeof(float) * d1);
    float *barycentric = (float *)malloc(sizeof(float) * (d + 2));
    short *rank = (short *)malloc(sizeof(short) * d1);
    short *canonical = (short *)malloc(sizeof(short) * d1 * d1);
    short *key = (short *)malloc(sizeof(short) * d1);

    for (int i = 0; i <= d; i++) {                                      /* :345-350 */
        for (int j = 0; j <= d - i; j++) canonical[i * d1 + j] = (short)i;
        for (int j = d - i + 1; j <= d; j++) canonical[i * d1 + j] = (short)(i - d1);
    }
    float inv_std_dev = (float)(sqrt(2.0 / 3.0) * (double)d1);          /* :353 */
    for (int i = 0; i < d; i++)                                         /* :355-356 */
        scale_factor[i] = (float)(1.0 / sqrt((double)((i + 2) * (i + 1))) * (double)inv_std_dev);

    for (int k = 0; k < N; k++) {
        const float *f = feature + (size_t)k * d;
        float sm = 0;                                                   /* :364-370 */
        for (int j = d; j > 0; j--) {
            float cf = f[j - 1] * scale_factor[j - 1];
            elevated[j] = sm - (float)j * cf;
            sm += cf;
        }
        elevated[0] = sm;

        float down_factor = 1.0f / (float)d1;                           /* :373-374 */
        float up_factor = (float)d1;
        int sum = 0;
        for (int i = 0; i <= d; i++) {                                  /* :376-391 */
            int rd2;
            float v = down_factor * elevated[i];
            float up = ceilf(v) * up_factor;
            float down = floorf(v) * up_factor;
            if (up - elevated[i] < elevated[i] - down) rd2 = (short)up;
            else rd2 = (short)down;
            rem0[i] = (float)rd2;
            sum = (int)((float)sum + (float)rd2 * down_factor);        /* `sum += rd2*down_factor` with an int sum */
        }
        for (int i = 0; i <= d; i++) rank[i] = 0;                       /* :394-403 */
        for (int i = 0; i < d; i++) {
            double di = (double)(elevated[i] - rem0[i]);
            for (int j = i + 1; j <= d; j++)
                if (di < (double)(elevated[j] - rem0[j])) rank[i]++;
                else rank[j]++;
        }
        for (int i = 0; i <= d; i++) {                                  /* :406-416 */
            rank[i] = (short)(rank[i] + sum);
            if (rank[i] < 0) { rank[i] = (short)(rank[i] + d1); rem0[i] += (float)d1; }
            else if (rank[i] > d) { rank[i] = (short)(rank[i] - d1); rem0[i] -= (float)d1; }
        }
        for (int i = 0; i <= d + 1; i++) barycentric[i] = 0;            /* :419-426 */
        for (int i = 0; i <= d; i++) {
            float v = (elevated[i] - rem0[i]) * down_factor;
            barycentric[d - rank[i]] += v;
            barycentric[d - rank[i] + 1] -= v;
        }
        barycentric[0] = (float)((double)barycentric[0] + (1.0 + (double)barycentric[d + 1]));   /* :428, in double */

        for (int remainder = 0; remainder <= d; remainder++) {         /* :431-437 */
            for (int i = 0; i < d; i++)
                key[i] = (short)(rem0[i] + (float)canonical[remainder * d1 + rank[i]]);
            L->offset[(size_t)k * d1 + remainder] = orc_hash_find(&H, key, 1);
            L->bary[(size_t)k * d1 + remainder] = barycentric[remainder];
        }
    }
    const int M = H.filled;                                             /* :450-473 */
    L->M = M;
    L->n1 = (int *)malloc(sizeof(int) * (size_t)d1 * (M > 0 ? M : 1));
    L->n2 = (int *)malloc(sizeof(int) * (size_t)d1 * (M > 0 ? M : 1));
    short *n1 = (short *)malloc(sizeof(short) * d1);
    short *n2 = (short *)malloc(sizeof(short) * d1);
    for (int j = 0; j <= d; j++)
        for (int i = 0; i < M; i++) {
            const short *kk = H.keys + (size_t)i * d;
            for (int k = 0; k < d; k++) { n1[k] = (short)(kk[k] - 1); n2[k] = (short)(kk[k] + 1); }
            if (j < d) { n1[j] = (short)(kk[j] + d); n2[j] = (short)(kk[j] - d); }   /* j == d: a slot find() never reads */
            L->n1[(size_t)j * M + i] = orc_hash_find(&H, n1, 0);
            L->n2[(size_t)j * M + i] = orc_hash_find(&H, n2, 0);
        }
    L->keys = H.keys;
    free(H.table);
    free(n1); free(n2); free(scale_factor); free(elevated); free(rem0); free(barycentric); free(rank); free(canonical); free(key);
}

/* which formulation orc_kernel_init uses: 0 = the SSE path (what the reference's build runs; every parity test), 1 = the
 * scalar path (cross-check only) */
static int orc_lattice_path = 0;
ORC_API void orc_set_lattice_path(int scalar) { orc_lattice_path = scalar ? 1 : 0; }

/* Permutohedral::seqCompute (permutohedral.cpp:476-527), used when
 * value_size <= 2 (permutohedral.cpp:600-601).  Note the blur is evaluated in
 * double (the literal 0.5) and the slice multiplies (w*value)*alpha. */
static void orc_seq_compute(const orc_lattice *L, float *out, const float *in, int vs) {
    const int d = L->d, d1 = d + 1, M = L->M, N = L->N;
    float *values = (float *)calloc((size_t)(M + 2) * vs, sizeof(float));
    float *newv = (float *)calloc((size_t)(M + 2) * vs, sizeof(float));
    for (int i = 0; i < N; i++)
        for (int j = 0; j <= d; j++) {
            int o = L->offset[(size_t)i * d1 + j] + 1;
            float w = L->bary[(size_t)i * d1 + j];
            for (int k = 0; k < vs; k++) {
                float t = w * in[(size_t)i * vs + k];
                values[(size_t)o * vs + k] = values[(size_t)o * vs + k] + t;
            }
        }
    for (int j = 0; j <= d; j++) {
        for (int i = 0; i < M; i++) {
            const float *oldv = values + (size_t)(i + 1) * vs;
            float *nv = newv + (size_t)(i + 1) * vs;
            const float *a = values + (size_t)(L->n1[(size_t)j * M + i] + 1) * vs;
            const float *b = values + (size_t)(L->n2[(size_t)j * M + i] + 1) * vs;
            for (int k = 0; k < vs; k++) {
                float s = a[k] + b[k];
                nv[k] = (float)((double)oldv[k] + 0.5 * (double)s);
            }
        }
        float *t = values; values = newv; newv = t;
    }
    const float alpha = 1.0f / (1.0f + powf(2.0f, (float)-d));
    for (int i = 0; i < N; i++) {
        for (int k = 0; k < vs; k++) out[(size_t)i * vs + k] = 0.0f;
        for (int j = 0; j <= d; j++) {
            int o = L->offset[(size_t)i * d1 + j] + 1;
            float w = L->bary[(size_t)i * d1 + j];
            for (int k = 0; k < vs; k++) {
                float t = w * values[(size_t)o * vs + k];
                t = t * alpha;
                out[(size_t)i * vs + k] = out[(size_t)i * vs + k] + t;
            }
        }
    }
    free(values); free(newv);
}

/* Permutohedral::sseCompute (permutohedral.cpp:529-589), element-wise
 * identical float arithmetic; the 4-wide padding of the channel dimension has
 * no numerical effect and is not reproduced. */
static void orc_sse_compute(const orc_lattice *L, float *out, const float *in, int vs) {
    const int d = L->d, d1 = d + 1, M = L->M, N = L->N;
    float *values = (float *)calloc((size_t)(M + 2) * vs, sizeof(float));
    float *newv = (float *)calloc((size_t)(M + 2) * vs, sizeof(float));
    float *tmp = (float *)malloc(sizeof(float) * vs);
    for (int i = 0; i < N; i++) {                        /* splat :545-553 */
        memcpy(tmp, in + (size_t)i * vs, sizeof(float) * vs);
        for (int j = 0; j <= d; j++) {
            int o = L->offset[(size_t)i * d1 + j] + 1;
            float w = L->bary[(size_t)i * d1 + j];
            float *v = values + (size_t)o * vs;
            for (int k = 0; k < vs; k++) { float t = w * tmp[k]; v[k] = v[k] + t; }
        }
    }
    for (int j = 0; j <= d; j++) {                       /* blur :556-569 */
        for (int i = 0; i < M; i++) {
            const float *oldv = values + (size_t)(i + 1) * vs;
            float *nv = newv + (size_t)(i + 1) * vs;
            const float *a = values + (size_t)(L->n1[(size_t)j * M + i] + 1) * vs;
            const float *b = values + (size_t)(L->n2[(size_t)j * M + i] + 1) * vs;
            for (int k = 0; k < vs; k++) { float s = a[k] + b[k]; s = 0.5f * s; nv[k] = oldv[k] + s; }
        }
        float *t = values; values = newv; newv = t;
    }
    const float alpha = 1.0f / (1.0f + powf(2.0f, (float)-d));    /* :571 */
    for (int i = 0; i < N; i++) {                        /* slice :574-584 */
        for (int k = 0; k < vs; k++) tmp[k] = 0.0f;
        for (int j = 0; j <= d; j++) {
            int o = L->offset[(size_t)i * d1 + j] + 1;
            float w = L->bary[(size_t)i * d1 + j] * alpha;
            const float *v = values + (size_t)o * vs;
            for (int k = 0; k < vs; k++) { float t = w * v[k]; tmp[k] = tmp[k] + t; }
        }
        memcpy(out + (size_t)i * vs, tmp, sizeof(float) * vs);
    }
    free(values); free(newv); free(tmp);
}

/* Permutohedral::compute dispatch (permutohedral.cpp:596-604). in/out are
 * vs x N column-major and may alias. */
static void orc_lattice_compute(const orc_lattice *L, float *out, const float *in, int vs) {
    if (vs <= 2) orc_seq_compute(L, out, in, vs);
    else orc_sse_compute(L, out, in, vs);
}

/* ------------------------------------------------------------------------- */
/* DenseKernel / PairwisePotential / DenseCRF — pairwise.cpp, densecrf.cpp     */
/* ------------------------------------------------------------------------- */

typedef struct {
    orc_lattice lat;
    float *norm;     /* [N] 1/sqrt(K1 + 1e-20)   (pairwise.cpp:44,54-57) */
    float w;         /* Potts weight             (labelcompatibility.cpp:46-48) */
} orc_kernel;

static void orc_kernel_init(orc_kernel *K, const float *feature, int d, int N, float w) {
    if (orc_lattice_path) orc_lattice_init_scalar(&K->lat, feature, d, N);
    else orc_lattice_init(&K->lat, feature, d, N);
    K->w = w;
    K->norm = (float *)malloc(sizeof(float) * N);
    float *ones = (float *)malloc(sizeof(float) * N);
    for (int i = 0; i < N; i++) ones[i] = 1.0f;
    orc_lattice_compute(&K->lat, K->norm, ones, 1);      /* pairwise.cpp:44 */
    for (int i = 0; i < N; i++)                          /* NORMALIZE_SYMMETRIC :54-57 */
        K->norm[i] = (float)(1.0 / sqrt((double)K->norm[i] + 1e-20));
    free(ones);
}
static void orc_kernel_free(orc_kernel *K) { orc_lattice_free(&K->lat); free(K->norm); K->norm = 0; }

/* PairwisePotential::apply (pairwise.cpp:173-178) = DenseKernel::filter
 * (pairwise.cpp:63-80, NORMALIZE_SYMMETRIC) followed by Potts (out = -w*out). */
static void orc_kernel_apply(const orc_kernel *K, float *out, const float *Q, int M_labels) {
    const int N = K->lat.N;
    for (int i = 0; i < N; i++)
        for (int k = 0; k < M_labels; k++) out[(size_t)i * M_labels + k] = Q[(size_t)i * M_labels + k] * K->norm[i];
    orc_lattice_compute(&K->lat, out, out, M_labels);
    const float nw = -K->w;
    for (int i = 0; i < N; i++)
        for (int k = 0; k < M_labels; k++) {
            float t = out[(size_t)i * M_labels + k] * K->norm[i];
            out[(size_t)i * M_labels + k] = nw * t;
        }
}

typedef struct {
    int W, H, M;               /* DenseCRFWrapper(W,H,nlabels) densecrf_wrapper.cpp:5-8 */
    float *unary;              /* [N*M] label-fastest = Eigen M x N column-major (:32-37) */
    orc_kernel kern[2];
    int nkern;
} orc_crf;

ORC_API orc_crf *orc_crf_create(int W, int H, int M) {
    orc_crf *c = (orc_crf *)calloc(1, sizeof(orc_crf));
    c->W = W; c->H = H; c->M = M;
    return c;
}
ORC_API void orc_crf_destroy(orc_crf *c) {
    if (!c) return;
    for (int k = 0; k < c->nkern; k++) orc_kernel_free(&c->kern[k]);
    free(c->unary); free(c);
}
ORC_API void orc_crf_set_unary_energy(orc_crf *c, const float *u) {
    size_t n = (size_t)c->W * c->H * c->M;
    if (!c->unary) c->unary = (float *)malloc(sizeof(float) * n);
    memcpy(c->unary, u, sizeof(float) * n);
}
/* DenseCRFWrapper::add_pairwise_energy (densecrf_wrapper.cpp:19-30): the
 * Gaussian potential (w2, theta_gamma) is added FIRST, then the bilateral one
 * (w1, theta_alpha, theta_beta).  Features per densecrf.cpp:61-81. */
ORC_API void orc_crf_add_pairwise_energy(orc_crf *c, float w1, float ta1, float ta2,
                                         float tb1, float tb2, float tb3, float w2,
                                         float tg1, float tg2, const unsigned char *im) {
    const int W = c->W, H = c->H, N = W * H;
    for (int k = 0; k < c->nkern; k++) orc_kernel_free(&c->kern[k]);
    float *f2 = (float *)calloc((size_t)2 * N, sizeof(float));
    for (int j = 0; j < H; j++)
        for (int i = 0; i < W; i++) {
            f2[(size_t)(j * W + i) * 2 + 0] = (float)i / tg1;
            f2[(size_t)(j * W + i) * 2 + 1] = (float)j / tg2;
        }
    orc_kernel_init(&c->kern[0], f2, 2, N, w2);
    free(f2);
    float *f5 = (float *)malloc(sizeof(float) * 5 * N);
    for (int j = 0; j < H; j++)
        for (int i = 0; i < W; i++) {
            size_t p = (size_t)(j * W + i);
            f5[p * 5 + 0] = (float)i / ta1;
            f5[p * 5 + 1] = (float)j / ta2;
            f5[p * 5 + 2] = (float)im[p * 3 + 0] / tb1;
            f5[p * 5 + 3] = (float)im[p * 3 + 1] / tb2;
            f5[p * 5 + 4] = (float)im[p * 3 + 2] / tb3;
        }
    orc_kernel_init(&c->kern[1], f5, 5, N, w1);
    free(f5);
    c->nkern = 2;
}

/* expAndNormalize (densecrf.cpp:98-106): per column subtract max, exp, divide
 * by the sum.  Eigen's vectorised exp / reduction order are not reproducible
 * without Eigen (unpinned at ulp level); expf and a left-to-right sum here. */
static void orc_exp_and_normalize(float *out, const float *in, int M, int N) {
    for (int i = 0; i < N; i++) {
        const float *b = in + (size_t)i * M;
        float *o = out + (size_t)i * M;
        float mx = b[0];
        for (int k = 1; k < M; k++) if (b[k] > mx) mx = b[k];
        float s = 0.0f;
        for (int k = 0; k < M; k++) { o[k] = expf(b[k] - mx); s = s + o[k]; }
        for (int k = 0; k < M; k++) o[k] = o[k] / s;
    }
}

/* DenseCRF::inference (densecrf.cpp:115-131) + output transposition of
 * DenseCRFWrapper::inference (densecrf_wrapper.cpp:45-50): out[i*M + l]. */
ORC_API void orc_crf_inference(const orc_crf *c, int n_iters, float *out) {
    const int N = c->W * c->H, M = c->M;
    size_t n = (size_t)N * M;
    float *Q = (float *)malloc(sizeof(float) * n);
    float *tmp1 = (float *)malloc(sizeof(float) * n);
    float *tmp2 = (float *)malloc(sizeof(float) * n);
    float *negu = (float *)malloc(sizeof(float) * n);
    for (size_t i = 0; i < n; i++) negu[i] = c->unary ? -c->unary[i] : 0.0f;
    orc_exp_and_normalize(Q, negu, M, N);
    for (int it = 0; it < n_iters; it++) {
        memcpy(tmp1, negu, sizeof(float) * n);
        for (int k = 0; k < c->nkern; k++) {
            orc_kernel_apply(&c->kern[k], tmp2, Q, M);
            for (size_t i = 0; i < n; i++) tmp1[i] = tmp1[i] - tmp2[i];
        }
        orc_exp_and_normalize(Q, tmp1, M, N);
    }
    memcpy(out, Q, sizeof(float) * n);
    free(Q); free(tmp1); free(tmp2); free(negu);
}
/* DenseCRF::map (densecrf.cpp:132-141) — first maximum wins. */
ORC_API void orc_crf_map(const orc_crf *c, int n_iters, int *labels) {
    const int N = c->W * c->H, M = c->M;
    float *Q = (float *)malloc(sizeof(float) * (size_t)N * M);
    orc_crf_inference(c, n_iters, Q);
    for (int i = 0; i < N; i++) {
        int m = 0;
        for (int k = 1; k < M; k++) if (Q[(size_t)i * M + k] > Q[(size_t)i * M + m]) m = k;
        labels[i] = m;
    }
    free(Q);
}
/* lattice introspection for tests: kernel 0 = Gaussian, 1 = bilateral */
ORC_API int orc_crf_lattice_size(const orc_crf *c, int k) { return k < c->nkern ? c->kern[k].lat.M : -1; }
ORC_API void orc_crf_lattice_norm(const orc_crf *c, int k, float *out) {
    memcpy(out, c->kern[k].norm, sizeof(float) * (size_t)c->W * c->H);
}
/* keys [M*d] short, per-pixel ids [N*(d+1)] and weights [N*(d+1)] */
ORC_API void orc_crf_lattice_dump(const orc_crf *c, int k, short *keys, int *offset, float *bary) {
    const orc_lattice *L = &c->kern[k].lat;
    if (keys) memcpy(keys, L->keys, sizeof(short) * (size_t)L->M * L->d);
    if (offset) memcpy(offset, L->offset, sizeof(int) * (size_t)L->N * (L->d + 1));
    if (bary) memcpy(bary, L->bary, sizeof(float) * (size_t)L->N * (L->d + 1));
}
/* blur neighbours [(d+1)*M] each, -1 = none (permutohedral.cpp:303-318) */
ORC_API void orc_crf_lattice_neighbours(const orc_crf *c, int k, int *n1, int *n2) {
    const orc_lattice *L = &c->kern[k].lat;
    const size_t n = (size_t)(L->d + 1) * (size_t)L->M;
    if (n1) memcpy(n1, L->n1, sizeof(int) * n);
    if (n2) memcpy(n2, L->n2, sizeof(int) * n);
}
/* one application of kernel k's lattice filter (Permutohedral::compute) to a
 * vs x N column-major matrix; for known-answer tests of splat/blur/slice. */
ORC_API void orc_crf_lattice_filter(const orc_crf *c, int k, const float *in, float *out, int vs) {
    orc_lattice_compute(&c->kern[k].lat, out, in, vs);
}

/* one application of DenseKernel::filter (pairwise.cpp:63-80, NORMALIZE_SYMMETRIC): out = norm . K (norm . in), without
 * the label compatibility; in/out [N*vs] label-fastest.  orc_kernel_apply above is this followed by Potts. */
ORC_API void orc_crf_kernel_filter(const orc_crf *c, int k, const float *in, float *out, int vs) {
    const orc_kernel *K = &c->kern[k];
    const int N = K->lat.N;
    for (int i = 0; i < N; i++)
        for (int l = 0; l < vs; l++) out[(size_t)i * vs + l] = in[(size_t)i * vs + l] * K->norm[i];
    orc_lattice_compute(&K->lat, out, out, vs);
    for (int i = 0; i < N; i++)
        for (int l = 0; l < vs; l++) out[(size_t)i * vs + l] = out[(size_t)i * vs + l] * K->norm[i];
}

/* ------------------------------------------------------------------------- */
/* Layer glue — pylayers/pylayers/pylayers.py                                  */
/* ------------------------------------------------------------------------- */

/* numpy's pairwise float64 add-reduce over a contiguous run of n < 128
 * elements (8 partial sums, then the tail) — the order np.sum(result, axis=1)
 * uses for the label axis in pylayers.py:86,330. */
static double orc_np_sum(const double *a, int n) {
    if (n < 8) { double r = 0.0; for (int i = 0; i < n; i++) r += a[i]; return r; }
    double r[8];
    for (int k = 0; k < 8; k++) r[k] = a[k];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int k = 0; k < 8; k++) r[k] += a[i + k];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

/* image preparation of CRFLayer.forward / DSRGLayer.refinement
 * (pylayers.py:70-75,315-319): order-1 zoom with the (in-1)/(out-1) mapping,
 * + mean pixel, np.round (half-even, float64), then .astype('ubyte')
 * (CRF.py:32).  images: (3,Hi,Wi) f32 -> im: (H*W*3) u8, channel-fastest. */
static void orc_prepare_image(const float *img, int Hi, int Wi, int H, int W, unsigned char *im) {
    static const double mean_pixel[3] = {104.0, 117.0, 123.0};
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            for (int ch = 0; ch < 3; ch++) {
                const float *pl = img + (size_t)ch * Hi * Wi;
                double sy = H > 1 ? (double)y * (double)(Hi - 1) / (double)(H - 1) : 0.0;
                double sx = W > 1 ? (double)x * (double)(Wi - 1) / (double)(W - 1) : 0.0;
                int y0 = (int)floor(sy), x0 = (int)floor(sx);
                double fy = sy - y0, fx = sx - x0;
                int y1 = y0 + 1 < Hi ? y0 + 1 : y0, x1 = x0 + 1 < Wi ? x0 + 1 : x0;
                double v;
                if (fy == 0.0 && fx == 0.0) v = (double)pl[(size_t)y0 * Wi + x0];
                else {
                    /* scipy interpolates separably (rows, then columns) in double */
                    double a = (1.0 - fy) * (double)pl[(size_t)y0 * Wi + x0] + fy * (double)pl[(size_t)y1 * Wi + x0];
                    double b = (1.0 - fy) * (double)pl[(size_t)y0 * Wi + x1] + fy * (double)pl[(size_t)y1 * Wi + x1];
                    v = (1.0 - fx) * a + fx * b;
                }
                float vf = (float)v;                     /* zoom returns the input dtype (f32) */
                double r = nearbyint((double)vf + mean_pixel[ch]);
                im[((size_t)y * W + x) * 3 + ch] = (unsigned char)(long long)r;
            }
}

/* CRFLayer.forward (pylayers.py:63-88) and DSRGLayer.refinement (:310-331) in
 * one call (they compute the same thing):
 *   probs   (B,C,H,W) f32, clipped IN PLACE to >= 1e-4           (:67,312)
 *   images  (B,3,Hi,Wi) f32 mean-subtracted
 *   refined (B,C,H,W) f64 = clip+renormalised marginals          (:84-86,328-330)
 *   logq    (B,C,H,W) f32 = log(refined)                          (:88)   (may be NULL)
 * CRF parameters as krahenbuhl2013.CRF (CRF.py:25-35). */
ORC_API void orc_crf_refine_batch(int B, int C, int H, int W, float *probs, const float *images,
                                  int Hi, int Wi, double scale_factor, int maxiter,
                                  double *refined, float *logq) {
    const int N = H * W;
    const float min_prob = 0.0001f;
    unsigned char *im = (unsigned char *)malloc((size_t)N * 3);
    float *unary = (float *)malloc(sizeof(float) * (size_t)N * C);
    float *q = (float *)malloc(sizeof(float) * (size_t)N * C);
    double col[256];
    for (int b = 0; b < B; b++) {
        float *p = probs + (size_t)b * C * N;
        for (size_t i = 0; i < (size_t)C * N; i++) if (p[i] < min_prob) p[i] = min_prob;
        for (int i = 0; i < N; i++)
            for (int c = 0; c < C; c++) unary[(size_t)i * C + c] = -p[(size_t)c * N + i];   /* CRF.py:28 */
        orc_prepare_image(images + (size_t)b * 3 * Hi * Wi, Hi, Wi, H, W, im);
        orc_crf *crf = orc_crf_create(W, H, C);
        orc_crf_set_unary_energy(crf, unary);
        orc_crf_add_pairwise_energy(crf, 10.0f, (float)(80.0 / scale_factor), (float)(80.0 / scale_factor),
                                    13.0f, 13.0f, 13.0f, 3.0f,
                                    (float)(3.0 / scale_factor), (float)(3.0 / scale_factor), im);
        orc_crf_inference(crf, maxiter, q);
        orc_crf_destroy(crf);
        for (int i = 0; i < N; i++) {
            for (int c = 0; c < C; c++) {
                double v = (double)q[(size_t)i * C + c];
                col[c] = v < 0.0001 ? 0.0001 : v;        /* min_prob is a Python float (f64) */
            }
            double s = orc_np_sum(col, C);
            for (int c = 0; c < C; c++) {
                double r = col[c] / s;
                refined[((size_t)b * C + c) * N + i] = r;
                if (logq) logq[((size_t)b * C + c) * N + i] = (float)log(r);
            }
        }
    }
    free(im); free(unary); free(q);
}

/* CRFLayer.backward (pylayers.py:90-92): grad = (1 - result) * top.diff. */
ORC_API void orc_crf_layer_backward(size_t n, const double *refined, const float *top_diff, float *bottom_diff) {
    for (size_t i = 0; i < n; i++) bottom_diff[i] = (float)((1.0 - refined[i]) * (double)top_diff[i]);
}

/* ------------------------------------------------------------------------- */
/* Connected components + seeded region growing                               */
/* ------------------------------------------------------------------------- */

static int orc_uf_find(int *parent, int x) {
    while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; }
    return x;
}
static void orc_uf_union(int *parent, int *rnk, int x, int y) {   /* CC_labeling_8.py:66-78 */
    int xr = orc_uf_find(parent, x), yr = orc_uf_find(parent, y);
    if (xr == yr) return;
    if (rnk[xr] < rnk[yr]) parent[xr] = yr;
    else if (rnk[xr] > rnk[yr]) parent[yr] = xr;
    else { parent[yr] = xr; rnk[xr]++; }
}

/* CC_lab.connectedComponentLabel (CC_labeling_8.py:112-197): two-pass
 * 8-connectivity labelling of equal-valued regions (both the 0- and the
 * 1-regions get labels), W/N/NW/NE neighbours, union by rank.  labels_out
 * receives the representative label of each pixel.  The representative ids
 * depend on the union order, which is reproduced (same neighbour order). */
ORC_API void orc_cc_label8(const int *mat, int H, int W, int *labels_out) {
    int n = H * W;
    int *parent = (int *)malloc(sizeof(int) * n), *rnk = (int *)malloc(sizeof(int) * n);
    int next = 0;
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            int val = mat[i * W + j];
            int lab[4], nl = 0;
            /* order: west, north, north-west, north-east (CC_labeling_8.py:264-280) */
            if (j > 0 && mat[i * W + j - 1] == val) lab[nl++] = labels_out[i * W + j - 1];
            if (i > 0 && mat[(i - 1) * W + j] == val) lab[nl++] = labels_out[(i - 1) * W + j];
            if (i > 0 && j > 0 && mat[(i - 1) * W + j - 1] == val) lab[nl++] = labels_out[(i - 1) * W + j - 1];
            if (i > 0 && j < W - 1 && mat[(i - 1) * W + j + 1] == val) lab[nl++] = labels_out[(i - 1) * W + j + 1];
            if (nl == 0) {
                labels_out[i * W + j] = next; parent[next] = next; rnk[next] = 0; next++;
            } else if (nl == 1) {
                labels_out[i * W + j] = lab[0];
            } else {
                int mn = lab[0];
                for (int l = 1; l < nl; l++) if (lab[l] < mn) mn = lab[l];
                labels_out[i * W + j] = mn;
                for (int l = 0; l + 1 < nl; l++) orc_uf_union(parent, rnk, lab[l], lab[l + 1]);
            }
        }
    for (int p = 0; p < n; p++) labels_out[p] = orc_uf_find(parent, labels_out[p]);
    free(parent); free(rnk);
}

/* generate_seed_step (pylayers.py:237-275) for one image.
 *   labels (C) f32 0/1, seed (C,H,W) f32 0/1 — mutated in place and returned,
 *   refined (C,H,W) f64, th1 (background) and th2 (foreground) thresholds. */
ORC_API void orc_srg_grow(int C, int H, int W, const float *labels, float *seed,
                          const double *refined, double th1, double th2) {
    const int N = H * W;
    int *cls = (int *)malloc(sizeof(int) * C), ncls = 0;
    for (int c = 0; c < C; c++) if (labels[c] == 1.0f) cls[ncls++] = c;      /* :240 */
    double *label_map = (double *)calloc(N, sizeof(double));                  /* :246 */
    /* :248-250  label_map[y,x] = c+1 for every cue; np.where enumerates in
     * C order (c ascending), fancy assignment keeps the last write */
    for (int c = 0; c < C; c++)
        for (int p = 0; p < N; p++) if (seed[(size_t)c * N + p] > 0.0f) label_map[p] = c + 1;
    if (ncls > 0)
        for (int p = 0; p < N; p++) {                                         /* :242-243,251-257 */
            int best = 0; double v = refined[(size_t)cls[0] * N + p];
            for (int k = 1; k < ncls; k++) {
                double t = refined[(size_t)cls[k] * N + p];
                if (t > v) { v = t; best = k; }                               /* argmax: first max */
            }
            int c = cls[best];
            if (v > th2) {
                if (c != 0) label_map[p] = c + 1;
                else if (v > th1) label_map[p] = c + 1;
            }
        }
    int *mat = (int *)malloc(sizeof(int) * N), *lab = (int *)malloc(sizeof(int) * N);
    char *hc = (char *)malloc((size_t)N + 1);
    for (int k = 0; k < ncls; k++) {                                          /* :259-273 */
        int c = cls[k];
        for (int p = 0; p < N; p++) mat[p] = (label_map[p] == (double)(c + 1));
        orc_cc_label8(mat, H, W, lab);
        memset(hc, 0, (size_t)N + 1);
        for (int p = 0; p < N; p++) {
            if (mat[p] == 1 && seed[(size_t)c * N + p] == 1.0f) hc[lab[p]] = 1;
            else if (mat[p] == 1) {
                float s = 0.0f;                                               /* np.sum over a f32 column */
                for (int q = 0; q < C; q++) s += seed[(size_t)q * N + p];
                if (s == 1.0f) lab[p] = -1;
            }
        }
        for (int p = 0; p < N; p++) if (lab[p] >= 0 && hc[lab[p]]) seed[(size_t)c * N + p] = 1.0f;
    }
    free(cls); free(label_map); free(mat); free(lab); free(hc);
}

/* DSRGLayer.generate_seed (pylayers.py:333-344) over a batch: seeds_out = cues,
 * then generate_seed_step per image. labels is (B,1,1,C). */
ORC_API void orc_srg_grow_batch(int B, int C, int H, int W, const float *labels, const float *cues,
                                const double *refined, double th1, double th2, float *seeds_out) {
    size_t per = (size_t)C * H * W;
    memcpy(seeds_out, cues, sizeof(float) * per * B);
    for (int b = 0; b < B; b++)
        orc_srg_grow(C, H, W, labels + (size_t)b * C, seeds_out + per * b, refined + per * b, th1, th2);
}

/* ------------------------------------------------------------------------- */
/* Pointwise layers (Theano expressions restated in closed form)               */
/* ------------------------------------------------------------------------- */

/* SoftmaxLayer.forward (pylayers.py:30-36,46-47), fp32:
 *   s = softmax_c(x);  p = (s + 1e-4) / sum_c(s + 1e-4) */
ORC_API void orc_softmax_forward(int B, int C, int HW, const float *x, float *p) {
    float s[256];
    for (int b = 0; b < B; b++)
        for (int i = 0; i < HW; i++) {
            const float *xi = x + (size_t)b * C * HW + i;
            float mx = xi[0];
            for (int c = 1; c < C; c++) if (xi[(size_t)c * HW] > mx) mx = xi[(size_t)c * HW];
            float z = 0.0f;
            for (int c = 0; c < C; c++) { s[c] = expf(xi[(size_t)c * HW] - mx); z += s[c]; }
            float z2 = 0.0f;
            for (int c = 0; c < C; c++) { s[c] = s[c] / z + 0.0001f; z2 += s[c]; }
            for (int c = 0; c < C; c++) p[(size_t)b * C * HW + (size_t)c * HW + i] = s[c] / z2;
        }
}
/* SoftmaxLayer.backward (pylayers.py:38-41,49-51) = T.grad(sum(probs*g), preds):
 *   dx_j = s_j (g_j - sum_k s_k g_k) / Z,  Z = sum_c(s_c + 1e-4) */
ORC_API void orc_softmax_backward(int B, int C, int HW, const float *x, const float *g, float *dx) {
    double s[256];
    for (int b = 0; b < B; b++)
        for (int i = 0; i < HW; i++) {
            const float *xi = x + (size_t)b * C * HW + i;
            const float *gi = g + (size_t)b * C * HW + i;
            double mx = xi[0];
            for (int c = 1; c < C; c++) if (xi[(size_t)c * HW] > mx) mx = xi[(size_t)c * HW];
            double z = 0.0;
            for (int c = 0; c < C; c++) { s[c] = exp((double)xi[(size_t)c * HW] - mx); z += s[c]; }
            double Z = 0.0, sg = 0.0;
            for (int c = 0; c < C; c++) { s[c] /= z; Z += s[c] + 1e-4; sg += s[c] * (double)gi[(size_t)c * HW]; }
            for (int c = 0; c < C; c++)
                dx[(size_t)b * C * HW + (size_t)c * HW + i] = (float)(s[c] * ((double)gi[(size_t)c * HW] - sg) / Z);
        }
}

/* BalancedSeedLossLayer (pylayers.py:126-152). Returns the loss; grad may be NULL.
 *   L = -mean_n[ sum S0 log p0 / max(|S0|,1e-4) ] - mean_n[ sum S_fg log p_fg / max(|S_fg|,1e-4) ] */
ORC_API double orc_seed_loss(int B, int C, int HW, const float *p, const float *S, float *grad) {
    double loss = 0.0;
    for (int b = 0; b < B; b++) {
        const float *pb = p + (size_t)b * C * HW, *Sb = S + (size_t)b * C * HW;
        double cbg = 0.0, cfg = 0.0, lbg = 0.0, lfg = 0.0;
        for (int i = 0; i < HW; i++) { cbg += Sb[i]; lbg += (double)Sb[i] * log((double)pb[i]); }
        for (size_t i = HW; i < (size_t)C * HW; i++) { cfg += Sb[i]; lfg += (double)Sb[i] * log((double)pb[i]); }
        double dbg = cbg > 1e-4 ? cbg : 1e-4, dfg = cfg > 1e-4 ? cfg : 1e-4;
        loss += -(lbg / dbg) / B - (lfg / dfg) / B;
        if (grad) {
            float *gb = grad + (size_t)b * C * HW;
            for (int i = 0; i < HW; i++) gb[i] = (float)(-(double)Sb[i] / ((double)pb[i] * dbg * B));
            for (size_t i = HW; i < (size_t)C * HW; i++) gb[i] = (float)(-(double)Sb[i] / ((double)pb[i] * dfg * B));
        }
    }
    return loss;
}

/* SeedLossLayer (pylayers.py:94-118; not referenced by the seed_mc prototxts).
 *   L = -mean_n[ sum_{c,hw} S log p / sum_{c,hw} S ]       (no 1e-4 floor: an empty seed map divides by zero, as Theano does) */
ORC_API double orc_seed_loss_plain(int B, int C, int HW, const float *p, const float *S, float *grad) {
    double loss = 0.0;
    for (int b = 0; b < B; b++) {
        const float *pb = p + (size_t)b * C * HW, *Sb = S + (size_t)b * C * HW;
        double cnt = 0.0, l = 0.0;
        for (size_t i = 0; i < (size_t)C * HW; i++) { cnt += Sb[i]; l += (double)Sb[i] * log((double)pb[i]); }
        loss += -(l / cnt) / B;
        if (grad) {
            float *gb = grad + (size_t)b * C * HW;
            for (size_t i = 0; i < (size_t)C * HW; i++) gb[i] = (float)(-(double)Sb[i] / ((double)pb[i] * cnt * B));
        }
    }
    return loss;
}

/* ExpandLossLayer (pylayers.py:183-233; SEC's global weighted rank pooling, unused by seed_mc).
 *   probs (B,C,HW) with class 0 = background; stat (B,C) image-level labels (stat[:,0] is ignored: pylayers.py:193).
 *   For every plane: sort ascending, weights w_k = q^(HW-1-k) (largest value gets weight 1), pooled = sum_k sort_k w_k / Z;
 *   q = 0.996 for the C-1 foreground planes, 0.999 for the background plane.  present = stat > 0.5.
 *   L1 = -mean_b sum_{c present} log pooled_bc / n_present;  L2 = -mean_b sum_{c absent} log(1 - max_bc) / n_absent;
 *   L3 = -mean_b log pooled_b0.  Ties: the sort is stable in the pixel index (lower index = lower rank); the max passes
 *   its gradient to every tied pixel (Theano's eq-mask rule). */
typedef struct { float v; int i; } orc_vi;
static int orc_vi_cmp(const void *a, const void *b) {
    const orc_vi *x = (const orc_vi *)a, *y = (const orc_vi *)b;
    if (x->v < y->v) return -1;
    if (x->v > y->v) return 1;
    return (x->i > y->i) - (x->i < y->i);
}
ORC_API double orc_expand_loss(int B, int C, int HW, const float *p, const float *stat, double q_fg, double q_bg,
                               float *grad) {
    double *w_fg = (double *)malloc(sizeof(double) * HW), *w_bg = (double *)malloc(sizeof(double) * HW);
    orc_vi *tmp = (orc_vi *)malloc(sizeof(orc_vi) * HW);
    double z_fg = 0.0, z_bg = 0.0, loss = 0.0;
    for (int k = 0; k < HW; k++) { w_fg[k] = pow(q_fg, (double)(HW - 1 - k)); w_bg[k] = pow(q_bg, (double)(HW - 1 - k)); }
    for (int k = 0; k < HW; k++) { z_fg += w_fg[k]; z_bg += w_bg[k]; }
    if (grad) memset(grad, 0, sizeof(float) * (size_t)B * C * HW);
    for (int b = 0; b < B; b++) {
        double n_pres = 0.0, n_abs = 0.0;
        for (int c = 1; c < C; c++) { if (stat[(size_t)b * C + c] > 0.5f) n_pres += 1.0; else n_abs += 1.0; }
        for (int c = 0; c < C; c++) {
            const float *pl = p + ((size_t)b * C + c) * HW;
            float *gl = grad ? grad + ((size_t)b * C + c) * HW : 0;
            const double *w = c == 0 ? w_bg : w_fg;
            const double z = c == 0 ? z_bg : z_fg;
            const int present = c > 0 && stat[(size_t)b * C + c] > 0.5f;
            if (c == 0 || present) {
                for (int i = 0; i < HW; i++) { tmp[i].v = pl[i]; tmp[i].i = i; }
                qsort(tmp, HW, sizeof(orc_vi), orc_vi_cmp);
                double pooled = 0.0;
                for (int k = 0; k < HW; k++) pooled += ((double)tmp[k].v * w[k]) / z;
                const double coef = c == 0 ? 1.0 / B : 1.0 / (n_pres * B);
                loss += -log(pooled) * coef;
                if (gl) for (int k = 0; k < HW; k++) gl[tmp[k].i] = (float)(-coef / pooled * (w[k] / z));
            } else {
                float mx = pl[0];
                for (int i = 1; i < HW; i++) if (pl[i] > mx) mx = pl[i];
                const double coef = 1.0 / (n_abs * B);
                loss += -log(1.0 - (double)mx) * coef;
                if (gl) for (int i = 0; i < HW; i++) if (pl[i] == mx) gl[i] = (float)(coef / (1.0 - (double)mx));
            }
        }
    }
    free(w_fg); free(w_bg); free(tmp);
    return loss;
}

/* ConfusionMatrix.add / generateM (training/tools/evaluate.py:25-30,61-68): M[gt, pred] += 1 over the pixels whose ground
 * truth passes the rule (add: gt != 255; generateM: gt < nclass). */
ORC_API void orc_confusion_matrix(size_t n, const unsigned char *gt, const unsigned char *pred, int nclass, int rule_lt,
                                  double *M) {
    for (size_t i = 0; i < n; i++) {
        if (rule_lt ? gt[i] < nclass : gt[i] != 255) M[(size_t)gt[i] * nclass + pred[i]] += 1.0;
    }
}

/* ConstrainLossLayer (pylayers.py:160-180).
 *   q = exp(lq);  L = mean_{n,hw} sum_c q log clip(q/p, 0.05, 20)
 *   dL/dp  = -(q/p) 1[0.05 <= q/p <= 20] / (B HW)
 *   dL/dlq = q (log clip(q/p) + 1[in range]) / (B HW) */
ORC_API double orc_constrain_loss(int B, int C, int HW, const float *p, const float *lq,
                                  float *grad_p, float *grad_lq) {
    double loss = 0.0;
    size_t n = (size_t)B * C * HW;
    double inv = 1.0 / ((double)B * HW);
    for (size_t i = 0; i < n; i++) {
        double q = exp((double)lq[i]);
        double r = q / (double)p[i];
        int in = (r >= 0.05 && r <= 20.0);
        double rc = r < 0.05 ? 0.05 : (r > 20.0 ? 20.0 : r);
        loss += q * log(rc);
        if (grad_p) grad_p[i] = (float)(in ? -(q / (double)p[i]) * inv : 0.0);
        if (grad_lq) grad_lq[i] = (float)(q * (log(rc) + (in ? 1.0 : 0.0)) * inv);
    }
    return loss * inv;
}
