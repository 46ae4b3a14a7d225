"""MI355X-native traversability filter chain (surface normals -> slope -> step -> roughness ->
weighted combine -> circular footprint) behind a C-ABI (include/travgpu.h, libtravgpu.so).

Python is plumbing only (tests, bench, multi-GPU launch); the product is the HIP library and the
C++ plugin adapters under traversability_estimation_amd/plugins/.
"""
from . import capi  # noqa: F401
from .capi import Context, TeError, default_params  # noqa: F401
