"""mpgan_amd -- MI355X-native (gfx950) hot path of MPGAN / GAPT.

``mpgan_amd.mpgan`` and ``mpgan_amd.gapt`` mirror the reference's ``mpgan`` / ``gapt`` packages
(same classes, keywords, state-dict keys); ``install_as_reference_packages()`` registers them
under those names so ``setup_training.py`` / ``train.py`` / ``gen.py`` pick them up unchanged.
"""
import sys as _sys

from . import _lib, ops  # noqa: F401
from . import data, checkpoint, gen  # noqa: F401
from . import mpgan  # noqa: F401
from . import gapt  # noqa: F401
from .mpgan import LinearNet, MPLayer, MPNet, MPGenerator, MPDiscriminator  # noqa: F401
from .gapt import MAB, SAB, PMA, ISAB, GAPT_G, GAPT_D  # noqa: F401


def install_as_reference_packages():
    """Make ``import mpgan`` / ``import gapt`` resolve to the MI355X implementations."""
    _sys.modules["mpgan"] = mpgan
    _sys.modules["gapt"] = gapt
