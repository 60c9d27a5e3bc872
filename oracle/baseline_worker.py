#!/usr/bin/env python
"""One process of bench.py's image-parallel CPU baseline (test infrastructure, like the rest of oracle/): the CPU
restatement of the supervision path on this process's own synthetic images for a fixed time.  The reference fans the
images of a batch out over a multiprocessing.Pool in the same way (pylayers.py:301-303,341-342).
  python oracle/baseline_worker.py SEED SECONDS   ->   prints "<images done> <seconds>"   """
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsrg_amd import synthetic as S          # noqa: E402  (numpy/scipy only)
from oracle import oracle as O               # noqa: E402


def main(seed, seconds):
    b = S.make_batch(seed, 4)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        sl = slice(n % 4, n % 4 + 1)
        probs = O.softmax_forward(b["logits"][sl])
        refined, logq = O.crf_refine_batch(probs, b["images"][sl], 12.0, 10)
        seeds = O.srg_grow_batch(b["labels"][sl], b["cues"][sl], refined)
        _, g1 = O.seed_loss(probs, seeds)
        _, g2, g3 = O.constrain_loss(probs, logq)
        O.softmax_backward(b["logits"][sl], g1 + g2 + O.crf_layer_backward(refined, g3))
        n += 1
    print(n, time.perf_counter() - t0, flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]), float(sys.argv[2]))
