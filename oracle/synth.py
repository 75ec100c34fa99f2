"""Oracle: host (numpy) evaluation of the synthetic-weight hash.  TEST INFRASTRUCTURE ONLY.

Independent restatement of the definition in chatts_amd/synth.py's docstring / include/chatts_amd.h
(chatts_fill_hash); tests check the HIP fill kernel against it bit for bit.
"""
import numpy as np

_M = np.uint32


def _mix32(x):
    x = x.astype(np.uint32, copy=True)
    x ^= x >> _M(16)
    x *= _M(0x7FEB352D)
    x ^= x >> _M(15)
    x *= _M(0x846CA68B)
    x ^= x >> _M(16)
    return x


def _mix32_scalar(x):
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def tensor_key(seed, name):
    h = 0x811C9DC5
    for b in name.encode("utf-8"):
        h = ((h ^ b) * 0x01000193) & 0xFFFFFFFF
    return _mix32_scalar((seed * 0x9E3779B9 + h) & 0xFFFFFFFF)


def bf16_bits(key, base, shift, rows, cols, row0=0, col0=0, full_cols=None):
    """uint16 bf16 bit patterns of the [rows, cols] block starting at (row0, col0)."""
    full_cols = cols if full_cols is None else full_cols
    with np.errstate(over="ignore"):
        r = (np.arange(rows, dtype=np.uint64) + np.uint64(row0))[:, None]
        c = (np.arange(cols, dtype=np.uint64) + np.uint64(col0))[None, :]
        i = r * np.uint64(full_cols) + c
        lo = (i & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        hi = (i >> np.uint64(32)).astype(np.uint32)
        h = _mix32(lo ^ _mix32(_M(key) + hi * _M(0x85EBCA6B)))
    n = ((h & _M(255)).astype(np.int32) + ((h >> _M(8)) & _M(255)).astype(np.int32)
         + ((h >> _M(16)) & _M(255)).astype(np.int32) + (h >> _M(24)).astype(np.int32) - 510)
    v = (np.float32(base) + n.astype(np.float32) * np.float32(2.0 ** -shift)).astype(np.float32)
    u = v.view(np.uint32)
    u = u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))
    return (u >> np.uint32(16)).astype(np.uint16)


def materialize(spec, seed, chunk_rows=4096):
    """float32 array of a TensorSpec-like object (name, rows, cols, base, shift, is_1d)."""
    key = tensor_key(seed, spec.name)
    out = np.empty((spec.rows, spec.cols), dtype=np.float32)
    for r0 in range(0, spec.rows, chunk_rows):
        r1 = min(spec.rows, r0 + chunk_rows)
        bits = bf16_bits(key, spec.base, spec.shift, r1 - r0, spec.cols, r0, 0, spec.cols)
        out[r0:r1] = (bits.astype(np.uint32) << np.uint32(16)).view(np.float32)
    return out[0] if spec.is_1d else out


def state_dict(specs, seed, as_torch=True, strip_prefix=""):
    import torch
    sd = {}
    for s in specs:
        a = materialize(s, seed)
        name = s.name[len(strip_prefix):] if strip_prefix and s.name.startswith(strip_prefix) else s.name
        sd[name] = torch.from_numpy(a) if as_torch else a
    return sd
