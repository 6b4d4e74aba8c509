"""Drop-in for the reference's ``mpgan`` package (``mpgan/__init__.py:1-3``): the model classes, ``mask_manual``
and the ``augment`` submodule that the reference's ``train.py:7`` imports from it."""
from .model import LinearNet, MPLayer, MPNet, MPGenerator, MPDiscriminator  # noqa: F401
from .mask_utils import mask_manual  # noqa: F401
from . import augment  # noqa: F401
