"""Builds bindings/dsrg_crf_wrapper.pyx (the Cython replacement of the reference's CRF/krahenbuhl2013/wrapper.pyx)
against libdsrg_hip.so:   python bindings/setup_cython.py build_ext --build-lib OUT --build-temp TMP"""
import os

import numpy
from Cython.Build import cythonize
from setuptools import Extension, setup

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIBDIR = os.path.join(ROOT, "dsrg_amd")
ext = Extension("dsrg_crf_wrapper", sources=[os.path.join(HERE, "dsrg_crf_wrapper.pyx")],
                include_dirs=[os.path.join(ROOT, "include"), numpy.get_include()],
                library_dirs=[LIBDIR], libraries=["dsrg_hip"], runtime_library_dirs=[LIBDIR],
                define_macros=[("NPY_NO_DEPRECATED_API", "NPY_1_7_API_VERSION")])
setup(name="dsrg_crf_wrapper", ext_modules=cythonize([ext], language_level=3, build_dir=os.environ.get("DSRG_CYTHON_BUILD_DIR")),
      script_args=None)
