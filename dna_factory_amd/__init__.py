"""Importable alias of the `dna-factory_amd/` package directory.

The package directory carries the project's name, which contains a hyphen
and so cannot be an import name; this shim makes `import dna_factory_amd`
resolve submodules from it.
"""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "dna-factory_amd")
if not _os.path.isdir(_PKG_DIR):
    raise ImportError("package directory %r is missing" % _PKG_DIR)
__path__.append(_PKG_DIR)

PACKAGE_DIR = _PKG_DIR
