# cython: language_level=3
# The binding a maintainer of the reference would put in place of CRF/krahenbuhl2013/wrapper.pyx:5-60 — the same Python
# class `DenseCRF` (set_unary_energy / add_pairwise_energy / inference / map), bound to the C ABI of libdsrg_hip.so
# (include/dsrg_hip.h) instead of the C++ class DenseCRFWrapper.  Built by bindings/setup_cython.py; compiled and imported
# by tests/test_cython_binding.py (CPU: links and raises without a device; GPU: equals the ctypes path).
import numpy as np
cimport numpy as cnp

cdef extern from "dsrg_hip.h":
    ctypedef struct dsrg_crf_s
    ctypedef dsrg_crf_s* dsrg_crf_t
    int dsrg_crf_create(int W, int H, int nlabels, dsrg_crf_t* out)
    int dsrg_crf_destroy(dsrg_crf_t h)
    int dsrg_crf_set_unary_energy(dsrg_crf_t h, const float* unary)
    int dsrg_crf_add_pairwise_energy(dsrg_crf_t h, float w1, float ta1, float ta2, float tb1, float tb2, float tb3,
                                     float w2, float tg1, float tg2, const unsigned char* im)
    int dsrg_crf_inference(dsrg_crf_t h, int n_iters, float* out)
    int dsrg_crf_map(dsrg_crf_t h, int n_iters, int* out)
    int dsrg_crf_npixels(dsrg_crf_t h)
    int dsrg_crf_nlabels(dsrg_crf_t h)
    const char* dsrg_last_error()


cdef _fail():
    raise RuntimeError(dsrg_last_error().decode("utf-8", "replace"))


cdef class DenseCRF:
    cdef dsrg_crf_t h

    def __cinit__(self, int W, int H, int nlabels):
        self.h = NULL
        if dsrg_crf_create(W, H, nlabels, &self.h) != 0:
            _fail()

    def __dealloc__(self):
        if self.h != NULL:
            dsrg_crf_destroy(self.h)

    def set_unary_energy(self, float[::1] unary_costs):
        if unary_costs.shape[0] != dsrg_crf_npixels(self.h) * dsrg_crf_nlabels(self.h):
            raise ValueError("unary_costs must hold npixels*nlabels floats")
        if dsrg_crf_set_unary_energy(self.h, &unary_costs[0]) != 0:
            _fail()

    def add_pairwise_energy(self, float w1, float theta_alpha_1, float theta_alpha_2, float theta_betta_1,
                            float theta_betta_2, float theta_betta_3, float w2, float theta_gamma_1, float theta_gamma_2,
                            unsigned char[::1] im):
        if im.shape[0] != dsrg_crf_npixels(self.h) * 3:
            raise ValueError("im must hold npixels*3 bytes")
        if dsrg_crf_add_pairwise_energy(self.h, w1, theta_alpha_1, theta_alpha_2, theta_betta_1, theta_betta_2,
                                        theta_betta_3, w2, theta_gamma_1, theta_gamma_2, &im[0]) != 0:
            _fail()

    def inference(self, int n_iters=10):
        probs = np.empty(dsrg_crf_npixels(self.h) * dsrg_crf_nlabels(self.h), dtype=np.float32)
        cdef float[::1] v = probs
        if dsrg_crf_inference(self.h, n_iters, &v[0]) != 0:
            _fail()
        return probs

    def map(self, int n_iters=10):
        labels = np.empty(dsrg_crf_npixels(self.h), dtype=np.int32)
        cdef int[::1] v = labels
        if dsrg_crf_map(self.h, n_iters, &v[0]) != 0:
            _fail()
        return labels

    def npixels(self):
        return dsrg_crf_npixels(self.h)

    def nlabels(self):
        return dsrg_crf_nlabels(self.h)
