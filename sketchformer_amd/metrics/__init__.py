"""Slow-metric registry (metrics/__init__.py of the reference): every SlowMetric subclass with a `name`."""
import importlib
import os
import pkgutil

from ..core.metrics import SlowMetric

for _loader, _name, _ispkg in pkgutil.iter_modules([os.path.dirname(__file__)]):
    importlib.import_module('.' + _name, __package__)


def _all_subclasses(cls):
    out = []
    for c in cls.__subclasses__():
        out.append(c)
        out.extend(_all_subclasses(c))
    return out


metrics_by_name = {c.name: c for c in _all_subclasses(SlowMetric) if getattr(c, 'name', None)}


def build_metric_by_name(metric_name, params):
    return metrics_by_name[metric_name](params)
