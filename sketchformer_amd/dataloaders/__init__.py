"""Data-loader registry (dataloaders/__init__.py:9-17 of the reference): every BaseDataLoader subclass
with a ``name`` attribute is discoverable by that name."""
from ..core.data import BaseDataLoader
from . import distributed_stroke3, synthetic_stroke3  # noqa: F401


def _all():
    seen, stack = {}, list(BaseDataLoader.__subclasses__())
    while stack:
        c = stack.pop()
        stack.extend(c.__subclasses__())
        if hasattr(c, "name"):
            seen[c.name] = c
    return seen


def get_dataloader_by_name(name):
    try:
        return _all()[name]
    except KeyError:
        raise KeyError("unknown data loader %r (have: %s)" % (name, ", ".join(sorted(_all()))))
