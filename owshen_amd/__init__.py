"""owshen_amd -- MI355X (gfx950) native Groth16 prover path for Owshen-style withdraw proofs.

The package is a thin host-side mirror over ``libowshen_gpu.so`` (C ABI in
``include/owshen_gpu.h``).  There is NO CPU fallback: importing ``owshen_amd.api`` loads
the HIP library and every compute call requires a visible gfx950 device.
"""
__all__ = ["api"]
__version__ = "0.1.0"
