"""gvpm_amd: MI355X-native photon gather + gradient-domain shift for the gvpm integrator.

`gvpm_amd.abi`   ctypes mirror of include/gvpm_hip.h
`gvpm_amd.host`  synthetic hosts (libgvpm_host.so)
`gvpm_amd.hip`   binding of the HIP library (libgvpm_hip.so); fails loudly if missing
"""
from . import abi  # noqa: F401
