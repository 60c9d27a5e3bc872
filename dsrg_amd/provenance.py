"""Which kernel sources a counter file was collected against: bench.py replays HBM-traffic / LDS counters from profiles/*.json
(PMC passes cannot run inside the timed bench) and must not quote them for a kernel that has changed since.  The collection
tools stamp the SHA-256 of the sources that define the named kernel; bench.py recomputes it and drops the figure on a mismatch."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dsrg_amd", "csrc")

# the translation units (and shared headers) behind each quoted kernel
KERNEL_SOURCES = {
    "mf_filter_kernel": ("meanfield.hip", "common.h", "embed.h"),
    "srg": ("srg.hip", "common.h"),
    "fullres": ("lattice_large.hip", "common.h", "embed.h"),
    "conv_igemm": ("conv_igemm.hip", "common.h"),
}


def sources_sha256(kernel):
    h = hashlib.sha256()
    for name in KERNEL_SOURCES[kernel]:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()


def all_sources_sha256():
    return {k: sources_sha256(k) for k in KERNEL_SOURCES}


def check(record, kernel):
    """record: a loaded profiles/*.json -> dict(file-independent provenance) with `match` True only when the file carries the
    hash of this kernel's sources and it equals the tree's"""
    have = (record or {}).get("sources_sha256", {}).get(kernel)
    now = sources_sha256(kernel)
    return {"collected_at_commit": (record or {}).get("commit"), "kernel_sources_sha256": have, "tree_sha256": now,
            "match": bool(have) and have == now}
