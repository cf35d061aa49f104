from . import data, metrics, models  # noqa: F401
