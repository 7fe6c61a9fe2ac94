"""Host logic of the image pipeline (coponerf_amd/pipeline.py): batching of model inputs along dim 0 and slicing of the
batched get_z results back into per-pair views.  CPU only."""
import torch

from coponerf_amd.pipeline import _collate, _pair_slice


def _inp(seed, B=1):
    g = torch.Generator().manual_seed(seed)
    return {"context": {"rgb": torch.rand(B, 2, 8, 8, 3, generator=g), "intrinsics": torch.rand(B, 2, 4, 4, generator=g),
                        "cam2world": torch.rand(B, 2, 4, 4, generator=g)},
            "query": {"uv": torch.rand(B, 1, 5, 2, generator=g), "cam2world": torch.rand(B, 1, 4, 4, generator=g)}}


def test_collate_concatenates_every_leaf_along_the_batch():
    a, b, c = _inp(1), _inp(2, B=2), _inp(3)
    cat = _collate([a, b, c])
    assert cat["context"]["rgb"].shape == (4, 2, 8, 8, 3)
    assert torch.equal(cat["context"]["rgb"][1:3], b["context"]["rgb"])
    assert torch.equal(cat["query"]["uv"][3], c["query"]["uv"][0])
    assert set(cat) == {"context", "query"} and set(cat["query"]) == {"uv", "cam2world"}


def test_pair_slice_follows_the_2B_layout_of_get_z():
    B = 3
    z = [torch.arange(2 * B * 4.0).view(2 * B, 4), torch.arange(2 * B * 2.0).view(2 * B, 2)]     # (2B, ...) feature maps
    flow = tuple(torch.arange(B * 3.0).view(B, 3) + k for k in range(4))                           # (B, ...) flows
    z1 = _pair_slice(z, 1, 2, 2)
    assert isinstance(z1, list) and torch.equal(z1[0], z[0][2:4]) and torch.equal(z1[1], z[1][2:4])
    f12 = _pair_slice(flow, 1, 3, 1)
    assert isinstance(f12, tuple) and all(torch.equal(f12[k], flow[k][1:3]) for k in range(4))
    assert z1[0].data_ptr() == z[0][2:4].data_ptr()          # views, not copies
