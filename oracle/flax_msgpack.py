"""TEST INFRASTRUCTURE -- not part of the product path (only tests/ may import this).

Restatement of the msgpack layout of ``flax.serialization`` (flax 0.10.4, pinned in the reference's ``uv.lock``; the
package is NOT installed here and is not vendored under /root/reference), used by the reference at
``meanflow_audio_codec/trainers/utils.py:45-58`` (``serialization.to_bytes(state)`` / ``from_bytes``).
Published algorithm (flax/serialization.py): ``msgpack_serialize`` = chunk every array leaf larger than
``MAX_CHUNK_SIZE = 2**30`` bytes into ``{'__msgpack_chunked_array__': True, 'shape': {'0': ..}, 'chunks': {'0': ..}}``,
then ``msgpack.packb(tree, default=_msgpack_ext_pack, strict_types=True)`` where an ndarray becomes
``ExtType(1, packb((shape, dtype.name, arr.tobytes('C')), use_bin_type=True))`` and a numpy scalar ``ExtType(3, ...)``.

Parity unpinned: no reference test or fixture holds a serialized TrainState, and flax cannot be run here.  This
in-memory encoder is deliberately written against the ``msgpack`` package's own packer (not the streaming writer
of ``meanflow_audio_codec_amd/trainers/checkpoint.py``) so the two can be compared byte for byte.
"""
import msgpack
import numpy as np

MAX_CHUNK_SIZE = 2 ** 30


def _ndarray_to_bytes(arr) -> bytes:
    arr = np.asarray(arr)
    return msgpack.packb((arr.shape, arr.dtype.name, arr.tobytes("C")), use_bin_type=True)


def _ext_pack(x):
    if isinstance(x, np.ndarray):
        return msgpack.ExtType(1, _ndarray_to_bytes(x))
    if isinstance(x, np.generic):
        return msgpack.ExtType(3, _ndarray_to_bytes(np.asarray(x)))
    return x


def _chunk(arr, max_bytes):
    per = max(1, int(max_bytes / arr.dtype.itemsize))
    flat = arr.reshape(-1)
    chunks = [flat[i:i + per] for i in range(0, flat.size, per)]
    return {"__msgpack_chunked_array__": True,
            "shape": {str(i): int(d) for i, d in enumerate(arr.shape)},
            "chunks": {str(i): c for i, c in enumerate(chunks)}}


def _chunk_leaves(tree, max_bytes):
    if isinstance(tree, dict):
        return {k: _chunk_leaves(v, max_bytes) for k, v in tree.items()}
    if isinstance(tree, np.ndarray) and tree.size * tree.dtype.itemsize > max_bytes:
        return _chunk(tree, max_bytes)
    return tree


def msgpack_serialize(tree, max_chunk_bytes=MAX_CHUNK_SIZE) -> bytes:
    return msgpack.packb(_chunk_leaves(tree, max_chunk_bytes), default=_ext_pack, strict_types=True)


def _ext_unpack(code, data):
    if code in (1, 3):
        shape, name, buf = msgpack.unpackb(data, raw=True)
        a = np.frombuffer(buf, dtype=np.dtype(name.decode())).reshape(shape)
        return a if code == 1 else a[()]
    return msgpack.ExtType(code, data)


def _unchunk(tree):
    if isinstance(tree, dict):
        if tree.get("__msgpack_chunked_array__"):
            shape = tuple(tree["shape"][str(i)] for i in range(len(tree["shape"])))
            return np.concatenate([tree["chunks"][str(i)] for i in range(len(tree["chunks"]))]).reshape(shape)
        return {k: _unchunk(v) for k, v in tree.items()}
    return tree


def msgpack_restore(data: bytes):
    return _unchunk(msgpack.unpackb(data, ext_hook=_ext_unpack, raw=False))
