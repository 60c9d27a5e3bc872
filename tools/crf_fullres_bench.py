#!/usr/bin/env python
"""Time the full-resolution test-time CRF (training/tools/test-ms.py:106) on the GPU and the CPU oracle."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import krahenbuhl2013
from dsrg_amd import synthetic as S
from dsrg_amd.crf import DenseCRF
from oracle import oracle as O
for (H, W) in [(321, 321), (375, 500)]:
    rng = np.random.default_rng(0)
    img = S.make_images(rng, 1, size=max(H, W))[0, :, :H, :W] + S.MEAN_PIXEL[:, None, None]
    im = np.ascontiguousarray(np.transpose(img, (1, 2, 0))).astype(np.uint8)
    logits = S.make_logits(rng, 1, 21, H, W, sigma=20.0)
    un = np.log(np.maximum(np.transpose(O.softmax_forward(logits)[0], (1, 2, 0)), 1e-5))
    q = krahenbuhl2013.CRF(im, un, scale_factor=1.0)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): q = krahenbuhl2013.CRF(im, un, scale_factor=1.0)
    torch.cuda.synchronize(); gpu_ms = (time.perf_counter() - t) / 5 * 1e3
    c = DenseCRF(W, H, 21); c.set_unary_energy(-un.ravel()); c.add_pairwise_energy(10, 80, 80, 13, 13, 13, 3, 3, 3, im.ravel())
    c.inference(10)
    t = time.perf_counter()
    for _ in range(5): c.inference(10)
    inf_ms = (time.perf_counter() - t) / 5 * 1e3
    imd, und = torch.from_numpy(im).cuda(), torch.from_numpy(un.astype(np.float32)).cuda()
    qd = krahenbuhl2013.CRF_device(imd, und, scale_factor=1.0)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): qd = krahenbuhl2013.CRF_device(imd, und, scale_factor=1.0)
    torch.cuda.synchronize(); dev_ms = (time.perf_counter() - t) / 5 * 1e3
    assert np.array_equal(qd.cpu().numpy(), q), "device-pointer path differs from the host-pointer path"
    lab = krahenbuhl2013.CRF_device(imd, und, scale_factor=1.0, want="map")
    assert np.array_equal(lab.cpu().numpy(), q.argmax(2))
    t = time.perf_counter(); qo = O.CRF(im, un, scale_factor=1.0); cpu_ms = (time.perf_counter() - t) * 1e3
    print("%dx%d: GPU CRF() %.2f ms from numpy, %.2f ms device-resident (inference only %.2f ms), CPU oracle %.1f ms, M_gauss %d M_bil %d, max|dQ| %.2e, argmax agree %.5f" % (
        H, W, gpu_ms, dev_ms, inf_ms, cpu_ms, c.lattice_size(0), c.lattice_size(1), np.abs(q - qo).max(), (q.argmax(2) == qo.argmax(2)).mean()), flush=True)
