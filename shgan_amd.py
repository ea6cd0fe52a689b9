"""Import alias: the package directory is named ``sh-gan_amd`` (not a valid Python identifier), so
``import shgan_amd`` loads it under this name and replaces this stub in ``sys.modules``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'sh-gan_amd')
_spec = importlib.util.spec_from_file_location('shgan_amd', os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules['shgan_amd'] = _pkg
_spec.loader.exec_module(_pkg)
