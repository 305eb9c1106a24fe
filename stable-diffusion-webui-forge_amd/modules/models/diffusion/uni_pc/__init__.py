from . import uni_pc  # noqa: F401
