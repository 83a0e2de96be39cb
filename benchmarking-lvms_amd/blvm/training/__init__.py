from .annealers import CosineAnnealer  # noqa: F401
