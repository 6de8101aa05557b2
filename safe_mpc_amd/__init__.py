"""MI355X-native batched safe-MPC engine: host package.

Only what the RTI hot path needs: configuration + URDF reading, problem assembly, the ctypes
binding of the HIP engine (include/smpc.h) and host mirrors of the reference's controller / cost / safe-set
interfaces.  The compute lives in csrc/ (hand-written HIP for gfx950).
"""
__version__ = '0.1.0'
