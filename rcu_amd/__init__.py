"""Import alias for the package directory ``reliability-challenges-uncertainty_amd/`` (whose name
is not a valid Python identifier): ``import rcu_amd`` / ``from rcu_amd import steps``."""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                         'reliability-challenges-uncertainty_amd')
__path__ = [_PKG_DIR]
with open(_os.path.join(_PKG_DIR, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(_PKG_DIR, '__init__.py'), 'exec'))
del _f
