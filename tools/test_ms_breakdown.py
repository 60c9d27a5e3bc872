#!/usr/bin/env python
"""where the time of inference.predict_mask_ms goes (one 375x500 image): wall milliseconds per stage, device synchronised after each"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "0")
import numpy as np, torch
from dsrg_amd import synthetic as S, inference as I
from dsrg_amd.backbone import VGG16ASPP
from dsrg_amd.crf import CRF_device
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = VGG16ASPP().to(dev).to(memory_format=torch.channels_last).eval()
H, W = 375, 500
rng = np.random.default_rng(1)
img = S.make_images(rng, 1, size=500)[0, :, :H, :W] + S.MEAN_PIXEL[:, None, None]
img = np.ascontiguousarray(np.transpose(img, (1, 2, 0))[:, :, ::-1]).clip(0, 255).astype(np.uint8)
def T(f, n=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    for _ in range(3): I.predict_mask_ms(net, img, device=dev)
    for size in (241, 321, 401):
        ms, x = T(lambda: I.preprocess(img, size, dev)); print("preprocess %d: %.2f ms" % (size, ms))
        ms, sc = T(lambda: net(x).float()); print("forward %d: %.2f ms" % (size, ms))
        ms, z = T(lambda: I._zoom(sc, H, W)); print("zoom %d: %.2f ms" % (size, ms))
    ms, pr = T(lambda: I._probs_from_scores(z[0] * 3)); print("softmax+clip: %.2f ms" % ms)
    un = torch.log(pr).permute(1, 2, 0).contiguous()
    ms, it = T(lambda: torch.as_tensor(np.asarray(img).astype('ubyte'), device=dev)); print("image upload: %.2f ms" % ms)
    ms, m = T(lambda: CRF_device(it, un, scale_factor=1.0, want="map")); print("CRF_device map: %.2f ms" % ms)
    ms, m = T(lambda: m.cpu().numpy()); print("mask download: %.2f ms" % ms)
    ms, _ = T(lambda: I.predict_mask_ms(net, img, device=dev)); print("predict_mask_ms: %.2f ms" % ms)
# the same stages in the order predict_mask_ms runs them (shapes change from call to call), synchronised after each
import contextlib
def tick(name, t0):
    torch.cuda.synchronize(); t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t0) * 1e3; return t1
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    for rep in range(4):
        acc = {}
        torch.cuda.synchronize(); t = time.perf_counter()
        total = None
        for size in (241, 321, 401):
            x = I.preprocess(img, size, dev); t = tick("preprocess", t)
            sc = net(x); t = tick("forward", t)
            sc = sc.float(); t = tick("float", t)
            z = I._zoom(sc, H, W); t = tick("zoom", t)
            total = z if total is None else total + z; t = tick("sum", t)
        pr = I._probs_from_scores(total[0]); t = tick("softmax", t)
        un = torch.log(pr).permute(1, 2, 0).contiguous(); t = tick("log+permute", t)
        it = torch.as_tensor(np.asarray(img).astype('ubyte'), device=dev); t = tick("upload", t)
        m = CRF_device(it, un, scale_factor=1.0, want="map"); t = tick("crf", t)
        m = m.cpu().numpy().astype(np.int64); t = tick("download", t)
        print("rep %d:" % rep, {k: round(v, 2) for k, v in acc.items()})
import gc
gc.disable()
st0 = torch.cuda.memory_stats()
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    ts = []
    for rep in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        I.predict_mask_ms(net, img, device=dev)
        ts.append((time.perf_counter() - t0) * 1e3)
st1 = torch.cuda.memory_stats()
print("gc off, 12 calls ms:", [round(t, 1) for t in ts])
print("device allocs %d -> %d, frees %d -> %d, retries %d" % (st0["num_device_alloc"], st1["num_device_alloc"], st0["num_device_free"], st1["num_device_free"], st1["num_alloc_retries"]))
