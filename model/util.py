"""Alias of egtr_amd.util under the reference's module path (model/util.py)."""
from egtr_amd.util import *  # noqa: F401,F403
