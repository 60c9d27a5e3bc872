"""dsrg_amd — MI355X-native (gfx950) implementation of the DSRG per-iteration
supervision path: dense-CRF mean field on a permutohedral lattice, seeded region
growing, and the seed/constrain losses, behind the reference's own Python-layer
and krahenbuhl2013.CRF interfaces.  See DESIGN.md."""
__all__ = ["ops", "crf", "layers", "synthetic"]
