"""The Cython binding of INTEGRATION.md §2 (bindings/dsrg_crf_wrapper.pyx: the replacement of the reference's
CRF/krahenbuhl2013/wrapper.pyx:5-60) is compiled for real, linked against libdsrg_hip.so and imported.
CPU: it links, exposes the reference's class and methods, and fails loudly without a device.  GPU: it returns exactly
what the ctypes path returns."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def wrapper(tmp_path_factory):
    from dsrg_amd import _lib
    _lib.lib()                                           # the library must exist (there is no fallback)
    out = tmp_path_factory.mktemp("cython_binding")
    env = dict(os.environ, DSRG_CYTHON_BUILD_DIR=str(out / "cy"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bindings", "setup_cython.py"), "build_ext", "--build-lib", str(out),
                        "--build-temp", str(out / "tmp")], capture_output=True, text=True, cwd=str(out), env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    sys.path.insert(0, str(out))
    try:
        import torch  # noqa: F401  (libdsrg_hip.so binds to torch's HIP runtime, as dsrg_amd._lib does)
        return importlib.import_module("dsrg_crf_wrapper")
    finally:
        sys.path.remove(str(out))


def test_binding_links_and_mirrors_the_reference_class(wrapper):
    assert hasattr(wrapper, "DenseCRF")
    for m in ("set_unary_energy", "add_pairwise_energy", "inference", "map"):      # wrapper.pyx:20-60
        assert callable(getattr(wrapper.DenseCRF, m))
    from dsrg_amd import _lib
    if _lib.lib().dsrg_device_count() < 1:
        with pytest.raises(RuntimeError):                # no device: constructor reports the C ABI's error text
            wrapper.DenseCRF(8, 8, 3)


@pytest.mark.gpu
def test_binding_equals_ctypes_path(wrapper):
    from dsrg_amd.crf import DenseCRF
    from dsrg_amd import synthetic as S
    rng = np.random.default_rng(0)
    for (H, W, C, scale) in [(41, 41, 21, 12.0), (80, 90, 5, 1.0)]:
        im = np.ascontiguousarray(np.transpose(S.make_images(rng, 1, size=max(H, W))[0, :, :H, :W] + S.MEAN_PIXEL[:, None, None],
                                               (1, 2, 0))).astype(np.uint8)
        un = rng.standard_normal((H, W, C)).astype(np.float32)
        outs = []
        for cls in (wrapper.DenseCRF, DenseCRF):
            c = cls(W, H, C)
            c.set_unary_energy(np.ascontiguousarray(-un.ravel()))
            c.add_pairwise_energy(10, 80 / scale, 80 / scale, 13, 13, 13, 3, 3 / scale, 3 / scale, im.ravel())
            outs.append((np.asarray(c.inference(10)).copy(), np.asarray(c.map(10)).copy()))
            del c
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
