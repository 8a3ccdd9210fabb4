"""Building blocks of the Res16UNet family on the pointcontrast_amd.minkowski modules (conv / BN helpers, residual block)."""
