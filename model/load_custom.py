"""Alias of egtr_amd.load_custom under the reference's module path (model/load_custom.py)."""
from egtr_amd.load_custom import load_cuda_kernels, load_hip_kernels  # noqa: F401
