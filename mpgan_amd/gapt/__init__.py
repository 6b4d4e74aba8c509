from .model import LinearNet, MAB, SAB, PMA, ISAB, GAPT_G, GAPT_D, _attn_mask  # noqa: F401
