"""MipRayMarcher2 (reference: training/volumetric_rendering/ray_marcher.py:20-62).

On the MI355X path the marcher is fused into the per-ray render kernel (csrc/renderer.hip); this class only exists so
that code poking at `renderer.ray_marcher` keeps working.  Calling it directly is not part of the hot path."""
import torch


class MipRayMarcher2(torch.nn.Module):
    def forward(self, colors, densities, depths, rendering_options):
        raise NotImplementedError('ray marching is fused into ImportanceRenderer.forward on the gfx950 path')
