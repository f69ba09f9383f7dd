from .deepmind import dm_decoder  # noqa: F401
