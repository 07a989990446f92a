"""Drop-in for the ``dICP`` package (lisusdaniil/dICP, absent upstream): ``from dICP.ICP import ICP``."""
from .ICP import ICP  # noqa: F401
