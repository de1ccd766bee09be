"""skirt9_amd -- MI355X-native primary-emission photon-packet engine behind SKIRT 9's ski/FITS interface.

Sub-modules:
  host    ctypes binding of the host model layer (ski file -> pmc_scene -> FITS/SED output)
  engine  ctypes binding of the HIP engine (libpmc.so, include/pmc.h)
"""
__all__ = ["host", "engine"]
