"""Drop-in for the reference package `krahenbuhl2013` (CRF/krahenbuhl2013/CRF.py,
wrapper.pyx): `from krahenbuhl2013 import CRF` (pylayers.py:16, tools/test-ms.py)."""
from dsrg_amd.crf import CRF, DenseCRF, CRF_device  # noqa: F401
