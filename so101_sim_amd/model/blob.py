"""Flat binary "model blob": the only model format that crosses the C ABI (include/so101.h,
`so101_create(model_blob, blob_bytes, ...)`).

Layout (little endian):
    u32 magic 'S1MB' | u32 version | u32 real_bytes (4 or 8) | u32 n_entries
    n_entries x { char name[32]; u32 kind (0=i32, 1=real); u32 count; u64 byte_offset }
    payload, each array 16-byte aligned
The same writer emits the f32 blob the HIP library consumes and the f64 blob the CPU oracle reads,
so both see one set of compile-time constants.
"""
from __future__ import annotations

import struct

import numpy as np

MAGIC = 0x424D3153   # 'S1MB'
VERSION = 3


def pack(model: dict, real=np.float32) -> bytes:
    real = np.dtype(real)
    entries = []
    for name, val in model.items():
        arr = np.asarray(val)
        if arr.dtype.kind in "iub":
            entries.append((name, 0, arr.astype("<i4").ravel()))
        else:
            entries.append((name, 1, arr.astype(real.newbyteorder("<")).ravel()))
    header = 16 + 48 * len(entries)
    off = (header + 15) // 16 * 16
    table, chunks = b"", []
    for name, kind, arr in entries:
        nb = name.encode()
        if len(nb) > 31:
            raise ValueError(f"blob entry name too long: {name}")
        table += struct.pack("<32sIIQ", nb, kind, arr.size, off)
        raw = arr.tobytes()
        pad = (-len(raw)) % 16
        chunks.append(raw + b"\0" * pad)
        off += len(raw) + pad
    out = struct.pack("<IIII", MAGIC, VERSION, real.itemsize, len(entries)) + table
    out += b"\0" * ((-len(out)) % 16)
    return out + b"".join(chunks)


def unpack(blob: bytes) -> dict:
    magic, version, rb, n = struct.unpack_from("<IIII", blob, 0)
    if magic != MAGIC or version != VERSION:
        raise ValueError("not a so101 model blob (magic/version mismatch)")
    real = np.dtype("<f4" if rb == 4 else "<f8")
    out = {}
    for k in range(n):
        name, kind, count, off = struct.unpack_from("<32sIIQ", blob, 16 + 48 * k)
        dt = np.dtype("<i4") if kind == 0 else real
        out[name.rstrip(b"\0").decode()] = np.frombuffer(blob, dtype=dt, count=count, offset=off).copy()
    return out
