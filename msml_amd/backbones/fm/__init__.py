from .fmoperator import FMCnn, FMNone  # noqa: F401
