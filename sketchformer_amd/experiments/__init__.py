"""Experiment registry (experiments/__init__.py of the reference)."""
import importlib
import os
import pkgutil

from ..core.experiments import Experiment

for _loader, _name, _ispkg in pkgutil.iter_modules([os.path.dirname(__file__)]):
    importlib.import_module('.' + _name, __package__)

experiments_by_name = {c.name: c for c in Experiment.__subclasses__() if getattr(c, 'name', None)}


def get_experiment_by_name(name):
    return experiments_by_name[name]
