"""MI355X-native PSO latent discovery (hot path of ltronchin/Gan-discovery-PSO).

Sub-packages mirror the reference's module paths so that
``from src.pso import util_pso`` becomes ``from <this package>.pso import util_pso``.
All arithmetic runs in libgdpso.so (HIP, gfx950); there is no CPU fallback.
"""
__all__ = ["engine", "synth", "pso", "utils", "inverter"]
