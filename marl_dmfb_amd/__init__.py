"""MI355X-native vectorised DMFB/MEDA droplet-routing environments + VDN training loop.

The environment kernels are hand-written HIP for gfx950 behind a C ABI (include/*.h,
marl_dmfb_amd/csrc/); this package is the host side that mirrors the reference's Python
object protocol (env / RolloutWorker / Agents / VDN).  There is no CPU fallback: importing
an env class without the built HIP library raises.
"""
__version__ = '0.1.0'
