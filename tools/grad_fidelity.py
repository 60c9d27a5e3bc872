#!/usr/bin/env python
"""Does the bf16 train step compute the float32 step's gradients?  (round-4 review, item 2)  Legs and set-up: dsrg_amd/fidelity.py.
per parameter: cosine to the float32 gradient and relative L2 distance.   usage: grad_fidelity.py [B] [--loss-own]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsrg_amd.fidelity import LEGS, gradient_fidelity, kaiming_                      # noqa: F401  (kaiming_: tools/overfit_probe.py)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2
    r = gradient_fidelity(B, LEGS, own_loss="--loss-own" in sys.argv, log=print)
    print("%-22s %10s | " % ("parameter", "|g| fp32") + " | ".join("%-17s" % t for t in LEGS) + "   (cosine / relative distance to float32)")
    for n, norm in r["ref_norm"].items():
        print("%-22s %10.3e | " % (n, norm) + " | ".join("%-17s" % ("%.5f %.2e" % (r["cos"][t][n], r["rel"][t][n])) for t in LEGS))
    print("min cosine over parameters: " + "  ".join("%s %.5f" % (t, min(r["cos"][t].values())) for t in LEGS))
    print("cosine of the whole gradient: " + "  ".join("%s %.6f" % (t, r["cos_all"][t]) for t in LEGS))


if __name__ == "__main__":
    main()
