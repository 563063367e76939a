"""Model registry with the reference's shape (model/__init__.py:28-55): name -> trainer class.

Only the hot-path trainers are built in; they are imported lazily so that a missing optional
dependency of some other plugin can never break ``--model MF``.  A stock ColdRec model file that
subclasses ``BaseColdStartTrainer`` can be registered with ``register(name, cls)``.
"""
import importlib

_BUILTIN = {'MF': ('.MF', 'MF'), 'LightGCN': ('.LightGCN', 'LightGCN'), 'DropoutNet': ('.DropoutNet', 'DropoutNet')}


class _Registry(dict):
    def __missing__(self, name):
        if name not in _BUILTIN:
            raise KeyError(name)
        mod, cls = _BUILTIN[name]
        self[name] = getattr(importlib.import_module(mod, __name__), cls)
        return self[name]

    def get(self, name, default=None):
        try:
            return self[name]
        except KeyError:
            return default

    def keys(self):
        return sorted(set(_BUILTIN) | set(dict.keys(self)))


AVAILABLE_MODELS = _Registry()


def register(name, cls):
    AVAILABLE_MODELS[name] = cls
