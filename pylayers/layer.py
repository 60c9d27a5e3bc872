"""Drop-in for the reference module `pylayers.layer` (train-f.prototxt:9: module 'pylayers.layer')."""
from dsrg_amd.data import ImageSegDataLayer, BatchLoader, SimpleTransformer  # noqa: F401
