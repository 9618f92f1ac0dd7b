"""simpleslam_amd -- MI355X-native scan-to-map registration (the PCR hot path of SimpleSLAM).

Only what the path needs: csrc/ (HIP kernels + the C ABI of include/pcr_hip.h),
pcr.py (host mirror of the reference's PCR::PointCloudRegister plugin), synth.py
(deterministic synthetic clouds for tests and bench.py).
"""
from .pcr import (LoamRegister, NdtRegister, PcrError, PointCloudRegister, ScanContext, SubMap, VgicpRegister,  # noqa: F401
                  default_params, load_library, make_register)
