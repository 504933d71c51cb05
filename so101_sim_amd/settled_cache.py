"""On-disk format of the settled-state store.

The reference pays `initializers.PropPlacer(settle_physics=True)` - up to 1000 physics substeps with the arm held - on
every `env.reset()` (so100_hand_over.py:222-229).  In the kernels the result of placement + settle is a pure function
of (model, seed, global env id, episode index, mass scale, solver settings, build), so it can be computed once
(`so101_compute_settled`), written to a file and handed back on later runs (`so101_set_settled_store`): the resets of
the covered episodes become a copy, bit-identical to settling again.

File layout (little endian):
    8 bytes   magic  b"SO101SS1"
    u32       length of the JSON header
    JSON      {"key": {...what the entries depend on...}, "first_episode": F, "n_episodes": E, "n_envs": N,
               "arrays": [[name, dtype, shape], ...]}
    raw arrays in header order: qpos f32 [E][20][N], qvel f32 [E][18][N], warmstart f32 [E][18][N], flags i32 [E][N]
    32 bytes  sha256 of everything before it

A file whose key differs from the environment's is refused with the differing fields named; nothing is silently
re-used across models, seeds, shards or builds.
"""
from __future__ import annotations

import hashlib
import json
import struct

import numpy as np

MAGIC = b"SO101SS1"
ARRAYS = (("qpos", "<f4", 20), ("qvel", "<f4", 18), ("warmstart", "<f4", 18), ("flags", "<i4", 0))


class SettledCacheError(RuntimeError):
    pass


def array_shape(name: str, n_episodes: int, n_envs: int) -> tuple:
    rows = dict((a[0], a[2]) for a in ARRAYS)[name]
    return (n_episodes, rows, n_envs) if rows else (n_episodes, n_envs)


def write(path: str, key: dict, first_episode: int, arrays: dict) -> None:
    """arrays: name -> numpy array of the shapes above (all four required)."""
    n_episodes, n_envs = arrays["flags"].shape
    header = {"key": key, "first_episode": int(first_episode), "n_episodes": int(n_episodes), "n_envs": int(n_envs), "arrays": []}
    blobs = []
    for name, dt, _ in ARRAYS:
        a = np.ascontiguousarray(arrays[name], dtype=np.dtype(dt))
        if a.shape != array_shape(name, n_episodes, n_envs):
            raise SettledCacheError(f"{name}: shape {a.shape}, expected {array_shape(name, n_episodes, n_envs)}")
        header["arrays"].append([name, dt, list(a.shape)])
        blobs.append(a.tobytes())
    hj = json.dumps(header, sort_keys=True).encode()
    body = MAGIC + struct.pack("<I", len(hj)) + hj + b"".join(blobs)
    with open(path, "wb") as f:
        f.write(body + hashlib.sha256(body).digest())


def read(path: str, expect_key: dict | None = None):
    """Returns (header, arrays).  Raises SettledCacheError on a damaged file or, when expect_key is given, on any
    differing key field."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 44 or data[:8] != MAGIC:
        raise SettledCacheError(f"{path}: not a settled-state file")
    body, digest = data[:-32], data[-32:]
    if hashlib.sha256(body).digest() != digest:
        raise SettledCacheError(f"{path}: checksum mismatch (truncated or corrupted)")
    (hl,) = struct.unpack("<I", body[8:12])
    if 12 + hl > len(body):
        raise SettledCacheError(f"{path}: header length out of range")
    header = json.loads(body[12:12 + hl].decode())
    if expect_key is not None:
        diff = sorted(k for k in set(expect_key) | set(header["key"]) if expect_key.get(k) != header["key"].get(k))
        if diff:
            raise SettledCacheError(f"{path}: settled states were computed for a different setup; differing fields: "
                                    + ", ".join(f"{k} (file {header['key'].get(k)!r}, env {expect_key.get(k)!r})" for k in diff))
    off, arrays = 12 + hl, {}
    E, N = header["n_episodes"], header["n_envs"]
    if [a[0] for a in header["arrays"]] != [a[0] for a in ARRAYS]:
        raise SettledCacheError(f"{path}: unexpected array list")
    for name, dt, shape in header["arrays"]:
        if tuple(shape) != array_shape(name, E, N):
            raise SettledCacheError(f"{path}: {name} has shape {shape}")
        n = int(np.prod(shape)) * np.dtype(dt).itemsize
        if off + n > len(body):
            raise SettledCacheError(f"{path}: array {name} runs past the end of the file")
        arrays[name] = np.frombuffer(body, dtype=np.dtype(dt), count=int(np.prod(shape)), offset=off).reshape(shape).copy()
        off += n
    if off != len(body):
        raise SettledCacheError(f"{path}: {len(body) - off} trailing bytes")
    return header, arrays
