#!/usr/bin/env python
"""CPU model of the LDS bank conflicts of mf_filter_kernel's gathers on the bilateral lattice (no GPU needed): what the
reference's first-occurrence vertex numbering costs, and what a renumbering would buy (VERDICT r1 item 5 / DESIGN §9).

The lattice (keys, per-pixel corners, blur neighbours) comes from the oracle's Permutohedral::init restatement for the images of
the bench batch.  Access pattern of the kernel (meanfield.hip, CPW = 2: one 8-byte element per vertex): thread t owns the
vertices t + 1024 k; per blur axis a wave issues, for every k, two ds_read_b64 gathers cur[n1[v]], cur[n2[v]] ("no neighbour" =
the zero slot M); the slice gathers cur[corner r of pixel t + 1024 p].  Bank model (cdna_hip_programming.md §LDS): a
ds_read_b64 is served in two groups of 32 lanes, the bank of byte address a is (a / 4) % 64, equal addresses broadcast, every
further distinct address on a busy bank costs one more cycle — so a group costs max over the 32 bank pairs of the number of
distinct elements n with n % 32 equal.  1.0 = conflict-free.

  python tools/lds_conflict_model.py [images]   ->  mean cycles per 32-lane group, per numbering"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from dsrg_amd import synthetic as S  # noqa: E402


def group_cycles(elems):
    """elems: (groups, 32) element indices -> cycles per group"""
    out = np.empty(elems.shape[0])
    for g, row in enumerate(elems):
        u = np.unique(row)
        out[g] = np.bincount(u % 32, minlength=32).max()
    return out


def blur_cost(n1, n2, M, perm):
    """perm[old id] = new id (identity = the reference's numbering); sentinel M stays M"""
    D1 = n1.shape[0]
    inv = np.empty(M, np.int64)
    inv[perm] = np.arange(M)                               # inv[new id] = old id
    pm = np.append(perm, M)                                # old id (or M for "none") -> new id
    cyc = []
    VPT, WG = 10, 1024
    for j in range(D1):
        for nb in (n1[j], n2[j]):
            tab = np.where(nb >= 0, nb, M)                # old neighbour ids per old vertex id
            new_nb = np.full(VPT * WG, M, np.int64)
            new_nb[:M] = pm[tab[inv]]                      # thread slot = new vertex id
            cyc.append(group_cycles(new_nb.reshape(-1, 32)))
    return float(np.mean(np.concatenate(cyc)))


def slice_cost(off, M, perm, N):
    pm = np.append(perm, M)
    PPT, WG = 2, 1024
    cyc = []
    for r in range(off.shape[1]):
        e = np.full(PPT * WG, M, np.int64)
        e[:N] = pm[off[:, r]]
        cyc.append(group_cycles(e.reshape(-1, 32)))
    return float(np.mean(np.concatenate(cyc)))


def main(n_images=4):
    b = S.make_batch(1000, max(n_images, 1))
    img = b["images"] + S.MEAN_PIXEL[None, :, None, None]
    im_u8 = np.ascontiguousarray(np.transpose(img, (0, 2, 3, 1)))[:, ::8, ::8, :].astype(np.uint8)   # the 41x41 images the CRF sees
    H, W = im_u8.shape[1:3]
    scale = 12.0
    rows = []
    for i in range(n_images):
        oc = O.DenseCRF(W, H, 21)
        oc.add_pairwise_energy(10, 80 / scale, 80 / scale, 13, 13, 13, 3, 3 / scale, 3 / scale, im_u8[i].ravel())
        keys, off, bary = oc.lattice_dump(1)
        n1, n2 = oc.lattice_neighbours(1)
        M, N = keys.shape[0], H * W
        ident = np.arange(M)
        lex = np.empty(M, np.int64)
        lex[np.lexsort(keys.T[::-1])] = np.arange(M)       # new id = rank in lexicographic key order
        rng = np.random.default_rng(i)
        rnd = rng.permutation(M)
        row = {"M": M}
        for name, perm in (("reference ids", ident), ("lexicographic key order", lex), ("random", rnd)):
            row[name] = (blur_cost(n1, n2, M, perm), slice_cost(off, M, perm, N))
        rows.append(row)
        print("image %d: M = %d" % (i, M))
        for name in ("reference ids", "lexicographic key order", "random"):
            print("   %-26s blur gathers %.2f cycles / group, slice gathers %.2f" % ((name,) + row[name]))
    print("mean over %d images:" % n_images)
    for name in ("reference ids", "lexicographic key order", "random"):
        print("   %-26s blur %.2f   slice %.2f" % (name, np.mean([r[name][0] for r in rows]), np.mean([r[name][1] for r in rows])))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
