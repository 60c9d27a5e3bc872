#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's
own Python (/root/reference/pylayers/pylayers/{pylayers,CC_labeling_8}.py) in the
build container.  Only the produced arrays are committed; the reference source
does not travel.

What is genuine reference code here:
  * CC_labeling_8.CC_lab            -> cc_cases.npz        (pure numpy, no stubs)
  * pylayers.generate_seed_step     -> srg_cases.npz       (numpy + CC_lab)
  * pylayers.CRFLayer.forward/backward, DSRGLayer.forward (the Python glue:
    in-place clip, scipy zoom, transposes, float64 clip/renorm, log, Pool.map)
                                    -> layer_glue.npz
  * pylayers.AnnotationLayer.setup/reshape/forward on a synthetic cue dictionary
                                    -> annotation_cases.npz
What is NOT reference code: importing pylayers.py needs its module-level imports
to resolve, so `caffe`, `theano`, `cPickle`, `cv2` are injected as EMPTY stub
modules (none of them is called by the functions exercised), and
`krahenbuhl2013.CRF` — whose C++ needs Eigen3 and cannot be built in this image —
is served by oracle.oracle.CRF.  layer_glue.npz therefore pins the glue around
the CRF, not the CRF arithmetic ("oracle").

Run:  python tests/golden/make_golden.py          (needs /root/reference)
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/pylayers/pylayers"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def import_reference():
    import yaml
    from oracle import oracle as O
    caffe = types.ModuleType("caffe")
    caffe.Layer = object
    theano = types.ModuleType("theano")
    theano.tensor = types.ModuleType("theano.tensor")
    kr = types.ModuleType("krahenbuhl2013")
    kr.CRF = O.CRF
    for name, mod in [("caffe", caffe), ("theano", theano), ("theano.tensor", theano.tensor),
                      ("cPickle", types.ModuleType("cPickle")), ("cv2", types.ModuleType("cv2")),
                      ("krahenbuhl2013", kr)]:
        sys.modules[name] = mod
    sys.path.insert(0, REF)
    import CC_labeling_8
    import pylayers as P
    P.xrange = range                                        # py2 builtin
    P.yaml = types.SimpleNamespace(load=yaml.safe_load)     # PyYAML>=6 needs a Loader
    return P, CC_labeling_8


class Blob(object):
    def __init__(self, data):
        self.data = data
        self.diff = np.zeros_like(data)

    def reshape(self, *shape):
        if self.data.shape != tuple(shape):
            self.data = np.zeros(shape, dtype=np.float32)
            self.diff = np.zeros(shape, dtype=np.float32)


def canonical_partition(lab):
    """Relabel by first occurrence so partitions can be compared."""
    lab = np.asarray(lab)
    out = np.empty(lab.shape, dtype=np.int32)
    seen = {}
    for i, v in enumerate(lab.ravel()):
        out.ravel()[i] = seen.setdefault(int(v), len(seen))
    return out


# ----------------------------------------------------------------------------
def srg_case_inputs():
    """Yield (name, labels[C], seed[C,H,W] f32, refined[C,H,W] f64)."""
    from dsrg_amd import synthetic as S
    from oracle import oracle as O
    C = 21
    # (a) realistic 41x41 cases: refined = clip/renorm softmax of smooth logits,
    #     rounded through float16 so the fixture stays small (exact in float64)
    for s in range(6):
        rng = np.random.default_rng(100 + s)
        H, W = (41, 41) if s < 5 else (65, 65)
        logits = S.make_logits(rng, 1, C, H, W, gain=60.0)
        labels, cues = S.make_labels_cues(np.random.default_rng(200 + s), 3, C, H, W)
        labels, cues = labels[2, 0, 0], cues[2]               # b%3==2 -> has multi-cue pixels
        # softmax over the PRESENT classes only (absent ones get the 1e-4 floor), so
        # large areas cross the 0.85 / 0.99 thresholds like CRF-refined maps do
        pres = np.where(labels == 1)[0]
        z = logits[0, pres].astype(np.float64)
        e = np.exp(z - z.max(0, keepdims=True))
        p = np.full((C, H, W), 1e-4)
        p[pres] = np.maximum(e / e.sum(0, keepdims=True), 1e-4)
        p = p / p.sum(0, keepdims=True)
        refined = p.astype(np.float16).astype(np.float64)
        yield "smooth%d" % s, labels, cues, refined
    rng = np.random.default_rng(7)

    def blank(H, W, present):
        labels = np.zeros(C, np.float32)
        labels[list(present)] = 1
        return labels, np.zeros((C, H, W), np.float32), np.full((C, H, W), 1.0 / C)

    # (b) serpentine component: one seed at the head of a 1-pixel-wide snake
    H, W = 13, 12
    labels, seed, ref = blank(H, W, (0, 3))
    snake = np.zeros((H, W), bool)
    for r in range(0, H, 2):
        snake[r, :] = True
        if r + 1 < H:
            snake[r + 1, (W - 1) if (r // 2) % 2 == 0 else 0] = True
    ref[3][snake] = 0.9
    ref[0][~snake] = 0.5
    seed[3, 0, 0] = 1
    yield "serpentine", labels, seed, ref
    # (c) diagonal-only connectivity (checkerboard): 8- vs 4-connectivity
    H, W = 9, 9
    labels, seed, ref = blank(H, W, (0, 5, 7))
    yy, xx = np.mgrid[0:H, 0:W]
    cb = (yy + xx) % 2 == 0
    ref[5][cb] = 0.95
    ref[7][~cb] = 0.86
    seed[5, 4, 4] = 1
    seed[7, 0, 1] = 1
    yield "checker", labels, seed, ref
    # (d) thresholds hit exactly (strict >), background needs > th1
    H, W = 6, 8
    labels, seed, ref = blank(H, W, (0, 2))
    ref[2, 0, :] = 0.85                       # == th2: not taken
    ref[2, 1, :] = np.nextafter(0.85, 1.0)    # just above: taken
    ref[0, 2, :] = 0.99                       # bg == th1: not taken
    ref[0, 3, :] = np.nextafter(0.99, 1.0)    # bg just above th1: taken
    ref[0, 4, :] = 0.9                        # bg above th2 but below th1: not taken
    seed[2, 0, 0] = seed[2, 1, 0] = 1
    seed[0, 2, 0] = seed[0, 3, 0] = seed[0, 4, 0] = 1
    yield "thresholds", labels, seed, ref
    # (e) multi-cue pixels, cue of an ABSENT class, exclusion rule
    H, W = 10, 10
    labels, seed, ref = blank(H, W, (0, 4, 9))
    ref[4][:, :5] = 0.97
    ref[9][:, 5:] = 0.97
    seed[4, 2, 1:4] = 1
    seed[9, 2, 2:4] = 1        # class 9 cues inside class-4 territory (single other cue -> excluded)
    seed[9, 7, 7] = 1
    seed[4, 7, 6:9] = 1        # class 4 cue inside class-9 territory, overlapping the class-9 seed
    seed[12, 5, 5] = 1         # absent class cue: highest class wins the label map but is never grown
    seed[0, 5, 5] = 1
    yield "multicue", labels, seed, ref
    # (f) argmax ties: first present class wins
    H, W = 5, 5
    labels, seed, ref = blank(H, W, (0, 1, 2))
    ref[1][:] = 0.9
    ref[2][:] = 0.9
    seed[1, 0, 0] = 1
    seed[2, 4, 4] = 1
    yield "ties", labels, seed, ref
    # (g) all background, 1x1, single row
    labels, seed, ref = blank(7, 7, (0,))
    ref[0][:] = 0.995
    seed[0, 3, 3] = 1
    yield "allbg", labels, seed, ref
    labels, seed, ref = blank(1, 1, (0, 1))
    ref[1][:] = 0.9
    seed[1, 0, 0] = 1
    yield "one", labels, seed, ref
    labels, seed, ref = blank(1, 17, (0, 6))
    ref[6][0, 3:12] = 0.9
    seed[6, 0, 5] = 1
    yield "row", labels, seed, ref
    # (h) random dense noise maps, many tiny components
    for s in range(4):
        H, W = int(rng.integers(8, 24)), int(rng.integers(8, 24))
        pres = [0] + sorted(int(x) for x in rng.choice(np.arange(1, C), size=3, replace=False))
        labels, seed, ref = blank(H, W, pres)
        r = rng.random((C, H, W)) ** 6
        r[[c for c in range(C) if c not in pres]] *= 0.01
        win = rng.integers(0, len(pres), size=(H, W))
        for k, c in enumerate(pres):
            r[c][win == k] += rng.choice([0.0, 3.0, 30.0], size=int((win == k).sum()))
        ref = r / r.sum(0, keepdims=True)
        seed = (rng.random((C, H, W)) < 0.03).astype(np.float32)
        yield "noise%d" % s, labels, seed, ref
    # (i) larger and oddly shaped noise maps: training sizes, the 128-column limit of the row-mask path and beyond it,
    #     many present classes, dense and sparse cues (separate generator so that the cases above keep their bytes)
    rng2 = np.random.default_rng(11)
    shapes = [(41, 41), (41, 41), (65, 65), (2, 128), (128, 3), (33, 97), (3, 150), (140, 5), (17, 64), (64, 17), (41, 41), (9, 129)]
    for s, (H, W) in enumerate(shapes):
        npres = int(rng2.integers(1, 9))
        pres = [0] + sorted(int(x) for x in rng2.choice(np.arange(1, C), size=npres, replace=False))
        labels, seed, ref = blank(H, W, pres)
        r = rng2.random((C, H, W)) ** 4
        r[[c for c in range(C) if c not in pres]] *= 0.01
        blocks = max(1, int(rng2.integers(1, 6)))                      # coarse winner map -> larger components
        win = rng2.integers(0, len(pres), size=((H + blocks - 1) // blocks, (W + blocks - 1) // blocks))
        win = np.kron(win, np.ones((blocks, blocks), dtype=win.dtype))[:H, :W]
        for k, c in enumerate(pres):
            r[c][win == k] += rng2.choice([0.0, 5.0, 50.0], size=int((win == k).sum()), p=[0.15, 0.35, 0.5])
        ref = r / r.sum(0, keepdims=True)
        seed = (rng2.random((C, H, W)) < (0.002 if s % 2 else 0.02)).astype(np.float32)
        yield "wide%d" % s, labels, seed, ref


def main():
    P, CC = import_reference()
    from dsrg_amd import synthetic as S

    # ---- connected components ------------------------------------------------
    rng = np.random.default_rng(1)
    mats, parts = [], []
    for k in range(12):
        H, W = int(rng.integers(1, 20)), int(rng.integers(1, 20))
        mat = (rng.random((H, W)) < rng.choice([0.3, 0.5, 0.7])).astype(int)
        cc = CC.CC_lab(mat)
        cc.connectedComponentLabel()
        mats.append(mat.astype(np.uint8))
        parts.append(canonical_partition(np.array(cc.labels)))
    np.savez_compressed(os.path.join(HERE, "cc_cases.npz"),
                        n=len(mats), **{"mat%d" % i: m for i, m in enumerate(mats)},
                        **{"part%d" % i: p for i, p in enumerate(parts)})

    # ---- generate_seed_step --------------------------------------------------
    out = {}
    names = []
    grown_total = 0
    for name, labels, seed, refined in srg_case_inputs():
        res = P.generate_seed_step([labels.copy(), seed.copy(), refined.copy(), 0.99, 0.85])
        names.append(name)
        out[name + "_labels"] = labels.astype(np.uint8)
        out[name + "_seed"] = seed.astype(np.uint8)
        if name.startswith("smooth"):
            out[name + "_refined_f16"] = refined.astype(np.float16)
            assert np.array_equal(refined.astype(np.float16).astype(np.float64), refined)
        else:
            out[name + "_refined"] = refined
        out[name + "_out"] = res.astype(np.uint8)
        grown_total += int(res.sum() - seed.sum())
        print("srg case %-11s %s seeds %5d -> %5d" % (name, seed.shape[1:], int(seed.sum()), int(res.sum())))
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "srg_cases.npz"), **out)
    print("grown pixels over all cases:", grown_total)

    # ---- layer glue (reference Python around the oracle CRF) -------------------
    glue = {}
    # "grow": two VOC-shaped images whose logits are sharpened towards the image's present classes (+6 / -6), so that the
    # marginals clear the 0.85 / 0.99 thresholds over whole regions and DSRGLayer.forward really grows (100 -> ~1800 seed
    # pixels) through the reference's own Python — "voc" grows 9 pixels, "tiny" 37
    for tag, (B, C, H, W, size) in {"voc": (1, 21, 41, 41, 321), "tiny": (2, 5, 11, 11, 81), "grow": (2, 21, 41, 41, 321)}.items():
        batch = S.make_batch({"voc": 11, "tiny": 12, "grow": 13}[tag], B, C, H, W, size=size)
        from oracle import oracle as O
        if tag == "grow":
            present = batch["labels"].reshape(B, C)[:, :, None, None]
            batch["logits"] = (batch["logits"] + 6.0 * (present * 2 - 1)).astype(np.float32)
        probs = O.softmax_forward(batch["logits"])
        labels, cues, images = batch["labels"], batch["cues"], batch["images"]
        if tag == "tiny":   # labels/cues generator assumes 21 classes; redo for C=5
            labels, cues = S.make_labels_cues(np.random.default_rng(5), B, C, H, W)
        crf = P.CRFLayer()
        b_probs, b_im = Blob(probs.copy()), Blob(images.copy())
        top = Blob(np.zeros_like(probs))
        crf.setup([b_probs, b_im], [top])
        crf.reshape([b_probs, b_im], [top])
        crf.forward([b_probs, b_im], [top])
        top.diff[...] = np.random.default_rng(3).standard_normal(top.diff.shape).astype(np.float32)
        crf.backward([top], [True, False], [b_probs, b_im])
        dsrg = P.DSRGLayer()
        dsrg.param_str = "{'th1': 0.99, 'th2': 0.85}"
        d_probs = Blob(probs.copy())
        bottoms = [Blob(labels.copy()), d_probs, Blob(cues.copy()), Blob(images.copy())]
        dtop = Blob(np.zeros_like(probs))
        dsrg.setup(bottoms, [dtop])
        dsrg.reshape(bottoms, [dtop])
        dsrg.forward(bottoms, [dtop])
        dsrg.pool.close()
        assert np.array_equal(d_probs.data, b_probs.data)          # both clip in place
        glue[tag + "_logits"] = batch["logits"]
        glue[tag + "_images_u8"] = np.rint(images + S.MEAN_PIXEL[None, :, None, None]).astype(np.uint8)
        glue[tag + "_labels"] = labels.astype(np.uint8)
        glue[tag + "_cues"] = cues.astype(np.uint8)
        glue[tag + "_probs_clipped"] = b_probs.data.copy()
        glue[tag + "_refined"] = crf.result.copy()                  # float64 (B,C,H,W)
        glue[tag + "_logq"] = top.data.copy()
        glue[tag + "_top_diff"] = top.diff.copy()
        glue[tag + "_crf_bottom_diff"] = b_probs.diff.copy()
        glue[tag + "_seeds"] = dtop.data.astype(np.uint8)
        print("glue %s: clipped %d probs, seeds %d -> %d" % (
            tag, int((probs < 1e-4).sum()), int(cues.sum()), int(dtop.data.sum())))
    np.savez_compressed(os.path.join(HERE, "layer_glue.npz"), **glue)

    # ---- AnnotationLayer (pylayers.py:346-387): the reference class itself on a synthetic cue dictionary ----------
    # (the real localization_cues-sal.pickle is a Google-Drive download; cPickle.load / open are served by stand-ins that
    # hand the dictionary over, everything else — setup, reshape, forward, the np.random flip — is the reference's code)
    ann = {}
    rng = np.random.default_rng(21)
    ids = np.array([7, 12, 3, 40], dtype=np.float32)
    data = {}
    for i in ids.astype(int):
        data['%i_labels' % i] = np.array(sorted(rng.choice(np.arange(1, 21), size=int(rng.integers(1, 4)), replace=False)))
        k = int(rng.integers(20, 60))
        data['%i_cues' % i] = np.stack([rng.integers(0, 21, k), rng.integers(0, 41, k), rng.integers(0, 41, k)])
        ann['%i_labels' % i], ann['%i_cues' % i] = data['%i_labels' % i], data['%i_cues' % i]
    images = rng.standard_normal((len(ids), 3, 9, 13)).astype(np.float32)
    sys.modules["cPickle"].load = lambda f: data
    P.open = lambda *a, **k: None
    for mirror in (False, True):
        lay = P.AnnotationLayer()
        lay.param_str = "{'cues': 'synthetic.pickle', 'mirror': %s}" % mirror
        bottoms = [Blob(ids.copy()), Blob(images.copy())]
        tops = [Blob(np.zeros(0, np.float32)), Blob(np.zeros(0, np.float32)), Blob(np.zeros(0, np.float32))]
        lay.setup(bottoms, tops)
        lay.reshape(bottoms, tops)
        np.random.seed(1234)
        lay.forward(bottoms, tops)
        tag = "mirror" if mirror else "plain"
        ann[tag + "_labels"], ann[tag + "_cues"], ann[tag + "_images"] = (tops[0].data.astype(np.uint8), tops[1].data.astype(np.uint8),
                                                                         tops[2].data.copy())
    del P.open
    ann["ids"], ann["images"], ann["seed"] = ids, images, np.array(1234)
    assert not np.array_equal(ann["mirror_cues"], ann["plain_cues"])
    np.savez_compressed(os.path.join(HERE, "annotation_cases.npz"), **ann)
    print("annotation layer: %d images, %d cue pixels" % (len(ids), int(ann["plain_cues"].sum())))
    # (pylayers/pylayers/layer.py — ImageSegDataLayer / BatchLoader / SimpleTransformer of train-f — is Python 2 only: the
    # print statements at layer.py:96,248-251 are syntax errors under this image's Python 3.10, so the module cannot be
    # imported and no fixture can be captured from it; dsrg_amd/data.py restates it and is tested against the statement-by-
    # statement expectations in tests/test_retrain_data.py.)
    for f in ("cc_cases.npz", "srg_cases.npz", "layer_glue.npz", "annotation_cases.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
