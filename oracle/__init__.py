"""CPU oracle for the MPGAN / GAPT hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``mpgan_amd/`` (the product) may import this
package.  It is imported by ``tests/``, by ``__graft_entry__.smoke()`` and by
``bench.py``'s ``cpu_baseline`` leg, always as the *checker* / the timed CPU baseline,
never as the thing shipped.

It is a plain-PyTorch (CPU, fp32 or fp64) functional restatement of the reference's
algorithm, written from the closed forms in SURVEY.md appendix A.3 / A.4, operating on
a flat ``state_dict`` whose key names are the reference's (so golden vectors captured
from the imported reference load directly).

Parity pin: every function here is checked by ``tests/test_oracle_golden.py`` against
``tests/golden/*.npz``, which were produced by ``tests/gen_golden.py`` importing the
reference's own ``mpgan`` / ``gapt`` packages in the build container (the reference has
no tests or golden vectors of its own -- SURVEY.md section 4).

The reference is Python only: there is nothing to compile into oracle/_ref/; the imported
reference itself played that role when the goldens were generated.
"""

from .mpgan_ref import (  # noqa: F401
    leaky,
    linearnet_forward,
    mplayer_forward,
    mpgen_forward,
    mpdisc_forward,
    gen_mask_from_labels,
    MPGAN_DEFAULTS,
)
from .gapt_ref import (  # noqa: F401
    mab_forward,
    sab_forward,
    pma_forward,
    isab_forward,
    gapt_g_forward,
    gapt_d_forward,
)
from .train_ref import (  # noqa: F401
    rmsprop_step,
    train_iteration,
    synthetic_batch,
    init_state_dict,
    mpgan_param_shapes,
    gapt_param_shapes,
)
