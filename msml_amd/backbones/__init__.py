from .msml import MSML  # noqa: F401
