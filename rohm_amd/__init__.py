"""rohm_amd: RoHM's iterative-denoising hot path on AMD Instinct MI355X (gfx950).

Python mirror of the reference's module layout over the C-ABI HIP library `librohm_hip.so`
(include/rohm_hip.h).  See README.md / DESIGN.md / INTEGRATION.md.
"""
__version__ = '0.1.0'
