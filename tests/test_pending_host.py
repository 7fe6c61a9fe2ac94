"""PendingHostTensor (coponerf_amd/render.py): the caller's `pixel_val` while its device->host copy may still be in flight.
Every way of getting at the VALUES or the BYTES must wait for the copy first; metadata must not.  Runs on the CPU with a
stand-in for the copy's event (the GPU behaviour is covered by tests/test_gpu_refloop.py)."""
import copy
import pickle

import numpy as np
import torch

from coponerf_amd.render import PendingHostTensor


class FakeEvent:
    def __init__(self):
        self.waits = 0

    def synchronize(self):
        self.waits += 1


def pending(shape=(2, 3, 4, 2)):
    host = torch.arange(int(np.prod(shape)), dtype=torch.float32).reshape(shape)
    ev = FakeEvent()
    return PendingHostTensor.wrap(host, ev, src=torch.zeros(1)), ev, host


def test_metadata_does_not_wait():
    t, ev, host = pending()
    assert t.shape == host.shape and t.dtype == torch.float32 and t.device.type == "cpu" and t.dim() == 4
    assert t.size(0) == 2 and t.numel() == 48 and len(t) == 2 and t.is_contiguous() and not t.is_cuda
    assert t.cpu() is t
    assert ev.waits == 0


ACCESSORS = {
    "data_ptr": lambda t: t.data_ptr(),
    "untyped_storage": lambda t: t.untyped_storage().data_ptr(),
    "numpy": lambda t: t.numpy(),
    "np.asarray": lambda t: np.asarray(t),
    "__array__": lambda t: t.__array__(),
    "tolist": lambda t: t.tolist(),
    "item": lambda t: t[0, 0, 0, 0].item(),
    "index": lambda t: t[1],
    "arithmetic": lambda t: t + 1,
    "clone": lambda t: t.clone(),
    "to_double": lambda t: t.double(),
    "sum": lambda t: t.sum(),
    "dlpack": lambda t: torch.from_dlpack(t),
    "__dlpack__": lambda t: t.__dlpack__(),
    "deepcopy": lambda t: copy.deepcopy(t),
    "deepcopy_dict": lambda t: copy.deepcopy({"pixel_val": t})["pixel_val"],
    "pickle": lambda t: pickle.loads(pickle.dumps(t)),
    "repr": lambda t: repr(t),
    "torch.cat": lambda t: torch.cat([t, t], dim=-3),
    "torch.cat(axis=)": lambda t: torch.cat([t, t], axis=-3),
    "torch.stack": lambda t: torch.stack([t, t]),
    "plain": lambda t: t.plain(),
}


def test_every_value_access_waits_first():
    for name, fn in ACCESSORS.items():
        t, ev, host = pending()
        got = fn(t)
        assert ev.waits == 1, name                       # ... exactly once,
        fn(t)
        assert ev.waits == 1, name                       # ... and never again
        assert t.__dict__.get("_cpn_src") is None, name  # the device source is released with the wait
        if isinstance(got, torch.Tensor) and name not in ("index", "arithmetic", "sum", "item"):
            assert not isinstance(got, PendingHostTensor) or got.__dict__.get("_cpn_ready") is None, name


def test_values_and_join_equal_plain_tensor():
    t, ev, host = pending()
    assert torch.equal(copy.deepcopy(t), host) and type(copy.deepcopy(t)) is torch.Tensor
    a, ea, ha = pending((2, 5, 4, 2))
    b, eb, hb = pending((2, 7, 4, 2))
    for kw in ({"dim": -3}, {"axis": -3}, {"dim": 1}):
        a, ea, ha = pending((2, 5, 4, 2))
        b, eb, hb = pending((2, 7, 4, 2))
        j = torch.cat([a, b], **kw)
        assert torch.equal(j, torch.cat([ha, hb], dim=1)) and ea.waits == 1 and eb.waits == 1
    a, ea, ha = pending((2, 5, 4, 2))
    assert torch.equal(torch.cat([a, hb], dim=1), torch.cat([ha, hb], dim=1)) and ea.waits == 1   # mixed list: stock path
    assert np.array_equal(np.asarray(pending()[0]), host.numpy())
