"""Drop-in for the reference package `pylayers` (pylayers/pylayers/__init__.py:1,
`python_param { module: 'pylayers' }` in train-s.prototxt:32,753,766,786,799,810):
the same layer classes, backed by the MI355X kernels of dsrg_amd."""
from dsrg_amd.layers import (SoftmaxLayer, CRFLayer, DSRGLayer, BalancedSeedLossLayer,  # noqa: F401
                             ConstrainLossLayer, AnnotationLayer, min_prob,
                             SeedLossLayer, ExpandLossLayer, AnnotationLayerCOCO)
