from .model import LinearNet, MPLayer, MPNet, MPGenerator, MPDiscriminator  # noqa: F401
