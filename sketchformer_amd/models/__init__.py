"""Model registry (models/__init__.py:9-17 of the reference)."""
from ..core.models import BaseModel
from . import sketchformer  # noqa: F401


def _all():
    seen, stack = {}, list(BaseModel.__subclasses__())
    while stack:
        c = stack.pop()
        stack.extend(c.__subclasses__())
        if isinstance(getattr(c, "name", None), str):
            seen[c.name] = c
    return seen


def get_model_by_name(name):
    try:
        return _all()[name]
    except KeyError:
        raise KeyError("unknown model %r (have: %s)" % (name, ", ".join(sorted(_all()))))
