"""fidelityfusion_amd -- MI355X-native drop-in for the GP hot path of IceLab-X/FidelityFusion.

Modules mirror the reference's names so that `import fidelityfusion_amd.kernel as kernel`,
`from fidelityfusion_amd.cigp_v10 import cigp as GPR`, `import fidelityfusion_amd.gp_computation_pack as gp_pack`
replace the reference imports one for one; all of them sit on libffgp.so (include/ffgp.h) and need a gfx950 GPU.
"""
__version__ = "0.1.0"
