"""nerfstyle_amd: MI355X-native volume-rendering hot path of "Locally Stylized Neural Radiance
Fields" (hkust-vgd/nerfstyle) -- occupancy-grid ray march + compaction, interleaved hash-grid
encode, fused MFMA MLPs and alpha compositing as hand-written gfx950 HIP kernels behind the plain
C ABI of include/nsr.h, driven by Python host code that mirrors the reference's module API.

    from nerfstyle_amd import raymarching            # reference: import raymarching
    from nerfstyle_amd.gridencoder import GridEncoder  # reference: from gridencoder import GridEncoder
    from nerfstyle_amd.network import Network          # reference: tcnn.Network
    from nerfstyle_amd.style_nerf import StyleTCNerf   # reference: networks.style_nerf.StyleTCNerf
    from nerfstyle_amd.renderer import Renderer        # reference: renderer.Renderer
"""
__version__ = '0.1.0'
