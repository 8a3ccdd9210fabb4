"""CPU oracle for the PointContrast pre-training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pointcontrast_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker / the timed CPU baseline.

What it restates (reference = /root/reference, "pc/" = pretrain/pointcontrast/):
  * MinkowskiEngine 0.4.3 semantics used by the path --
    ``sparse_ref.py``.  MinkowskiEngine is a third-party dependency pinned only
    by prose (README.md:24,34 -> tag v0.4.3); its source is NOT under
    /root/reference and it is not installable here, so the ME-specific
    conventions (hash iteration order, kernel-offset enumeration) are
    **parity unpinned** against ME itself.  What IS pinned: sparse conv /
    strided conv / transposed conv semantics against dense
    ``torch.nn.functional.conv3d`` / ``conv_transpose3d`` (tests/test_oracle_dense.py),
    BatchNorm / CrossEntropy / SGD against torch's own implementations.
  * Network wiring pc/model/res16unet.py:17-275, pc/model/resnet.py:99-140,
    pc/model/modules/resnet_block.py:13-60 -- ``model_ref.py``.
  * Losses pc/lib/ddp_trainer.py:39-51,182-238,400-426, pc/lib/criterion.py:10-19
    -- ``loss_ref.py``.
"""
