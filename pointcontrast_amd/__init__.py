"""pointcontrast_amd -- MI355X-native (gfx950) sparse-voxel contrastive pre-training path.

  pointcontrast_amd.minkowski   the MinkowskiEngine 0.4.3 surface the path uses, on libpcmi
  pointcontrast_amd.model       Res16UNet14 / 34 / 34C
  pointcontrast_amd.lib         trainers, DDP reducer, samplers, synthetic pair generator
  pointcontrast_amd.functional  autograd wrappers over the C ABI (include/pcmi.h)
The HIP library is loaded lazily by the modules that need it; importing this package alone
does not require a GPU or the built library.
"""
__version__ = "0.1.0"
