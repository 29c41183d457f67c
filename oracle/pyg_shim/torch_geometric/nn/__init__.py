from . import inits, conv  # noqa: F401
