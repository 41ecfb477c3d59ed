"""Tensors across a multiprocessing queue BY VALUE.  torch's default reduction passes a CPU tensor's storage as a file descriptor that the RECEIVER
fetches from the sender's resource-sharer thread when it unpickles — if the worker has already exited (its job is done once the message is
flushed) the parent gets `ConnectionResetError` (seen once in round 6's final full-suite run, tests/test_gpu_vae_tiled.py).  `ship` turns every
tensor of a nested structure into (dtype, shape, bytes); `unship` rebuilds it."""
import torch

_TAG = "__tensor_by_value__"


def ship(obj):
    if isinstance(obj, torch.Tensor):
        t = obj.detach().cpu().contiguous()
        return (_TAG, str(t.dtype).replace("torch.", ""), tuple(t.shape), t.view(torch.uint8).numpy().tobytes() if t.numel() else b"")
    if isinstance(obj, (list, tuple)):
        return type(obj)(ship(o) for o in obj)
    if isinstance(obj, dict):
        return {k: ship(v) for k, v in obj.items()}
    return obj


def unship(obj):
    if isinstance(obj, tuple) and len(obj) == 4 and obj[0] == _TAG:
        _, dt, shape, raw = obj
        dtype = getattr(torch, dt)
        if not raw:
            return torch.empty(shape, dtype=dtype)
        return torch.frombuffer(bytearray(raw), dtype=dtype).reshape(shape).clone()
    if isinstance(obj, (list, tuple)):
        return type(obj)(unship(o) for o in obj)
    if isinstance(obj, dict):
        return {k: unship(v) for k, v in obj.items()}
    return obj
