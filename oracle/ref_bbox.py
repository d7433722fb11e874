"""Loader for oracle/_ref/bbox*.so: the reference's own Cython ``bbox_overlaps`` / ``bbox_intersections``
(lib/fpn/box_intersections_cpu/bbox.pyx:21-61 / 64-108) compiled by oracle/Makefile from the sources under
/root/reference.  TEST INFRASTRUCTURE ONLY (validates oracle/postprocess.py, generates tests/golden/postprocess.npz).
Returns None when the module has not been built (e.g. a checkout without /root/reference)."""
import glob
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    hits = glob.glob(os.path.join(_HERE, "_ref", "bbox*.so"))
    if not hits:
        return None
    import numpy as np
    if not hasattr(np, "float"):
        np.float = float  # the module-level ``DTYPE = np.float`` (bbox.pyx:12) predates numpy 1.24
    spec = importlib.util.spec_from_file_location("bbox", hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
