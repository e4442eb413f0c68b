"""fieldconv_amd -- MI355X-native field convolutions behind the FieldConv module API.

`from fieldconv_amd.nn import FieldConv, FCResNetBlock, ECHOBlock, TangentLin, TangentNonLin`
mirrors `from nn import ...` of the reference (reference nn/__init__.py:1-12).
"""
__version__ = '0.1.0'

from . import nn, transforms, utils  # noqa: F401,E402
from . import optim  # noqa: F401,E402
from ._lib import arithmetic  # noqa: F401,E402   (`with fieldconv_amd.arithmetic('f32'):` -- arithmetic mode of the convolutions launched inside)
