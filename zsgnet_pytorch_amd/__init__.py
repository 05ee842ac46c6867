"""Importable alias of the hyphenated package directory `zsgnet-pytorch_amd/` (a hyphen cannot appear in a Python
module name).  `import zsgnet_pytorch_amd.mdl` resolves to `zsgnet-pytorch_amd/mdl.py`."""
import os

__path__ = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zsgnet-pytorch_amd")]
with open(os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), os.path.join(__path__[0], "__init__.py"), "exec"))
