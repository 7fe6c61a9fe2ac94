"""coponerf_amd — MI355X-native (gfx950) implementation of CoPoNeRF's per-ray rendering hot path.

`from coponerf_amd import CoPoNeRF; CoPoNeRF.CoPoNeRF(n_view=2)` mirrors the reference's
`from models import CoPoNeRF; CoPoNeRF.CoPoNeRF(n_view=2)`.
"""
__all__ = ["CoPoNeRF", "render", "synthetic", "_hip"]
