"""Checkpoints in the reference's on-disk format -- SURVEY 8(f) row N2.

The reference writes ``flax.serialization.to_bytes(TrainState)`` to ``checkpoints/step_%05d.msgpack``
(``trainers/utils.py:45-58``) with a JSON sidecar (``:111-177``) and manages them with
``find_latest_checkpoint`` / ``validate_checkpoint`` / ``cleanup_old_checkpoints`` / ``load_checkpoint_and_resume``
(``:512-784``).  flax (0.10.4, ``uv.lock``) is not installed here, so its published msgpack layout is restated:

* the state dict of ``TrainState`` has the pytree fields ``step``, ``params`` and ``opt_state``; for
  ``optax.adamw`` (= ``chain(scale_by_adam, add_decayed_weights, scale_by_learning_rate)``) the optimizer state is
  a 3-tuple ``(ScaleByAdamState(count, mu, nu), EmptyState(), EmptyState())``, which flax serialises as
  ``{"0": {"count", "mu", "nu"}, "1": {}, "2": {}}`` (tuples -> dicts keyed by decimal index);
* every array leaf is msgpack ``ExtType(1, packb((shape, dtype_name, raw_bytes)))`` (numpy scalars: code 3);
* a leaf larger than 2**30 bytes is replaced by ``{"__msgpack_chunked_array__": True, "shape": {"0": ..},
  "chunks": {"0": <array>, "1": ...}}`` with chunks of ``2**30 // itemsize`` elements of the flattened array.

No reference test pins this format and flax cannot be run here: **parity unpinned** (the tests check the writer
against an independent in-memory encoder and the reader against both).

The literal config's state is 165 GB, so both directions stream: the writer emits msgpack headers itself and
copies one <= 1 GiB chunk at a time from the device; the reader walks the file and fills the template's tensors
leaf by leaf (nothing the size of the checkpoint is ever resident on the host).
"""
from __future__ import annotations

import hashlib
import json
import platform
import re
import struct
import subprocess
import sys
from datetime import datetime
from pathlib import Path
from typing import Any, BinaryIO, Callable, Iterator

import msgpack
import numpy as np
import torch

MAX_CHUNK_BYTES = 2 ** 30
EXT_NDARRAY, EXT_NPSCALAR = 1, 3
_CHUNK_KEY = "__msgpack_chunked_array__"

_TORCH_TO_NAME = {torch.float32: "float32", torch.float64: "float64", torch.float16: "float16",
                  torch.bfloat16: "bfloat16", torch.int32: "int32", torch.int64: "int64", torch.uint8: "uint8",
                  torch.int8: "int8", torch.int16: "int16", torch.bool: "bool"}
_NAME_TO_TORCH = {v: k for k, v in _TORCH_TO_NAME.items()}


# ---------------------------------------------------------------------------
# pytree <-> nested dict
# ---------------------------------------------------------------------------
def nest(flat: dict) -> dict:
    """{"blocks_0/input_proj1/kernel": t} -> {"blocks_0": {"input_proj1": {"kernel": t}}} (flax param tree)."""
    out: dict = {}
    for path, v in flat.items():
        d = out
        parts = path.split("/")
        for p in parts[:-1]:
            d = d.setdefault(p, {})
        d[parts[-1]] = v
    return out


def flatten(tree: dict, prefix: str = "") -> dict:
    out = {}
    for k, v in tree.items():
        path = f"{prefix}/{k}" if prefix else str(k)
        if isinstance(v, dict):
            out.update(flatten(v, path))
        else:
            out[path] = v
    return out


def state_dict(state) -> dict:
    """The dict ``flax.serialization.to_state_dict(TrainState)`` would produce for this state."""
    return {
        "step": int(state.step),
        "params": nest(state.params),
        "opt_state": {
            "0": {"count": np.asarray(int(state.step), dtype=np.int32),
                  "mu": nest(state.opt_state["mu"]), "nu": nest(state.opt_state["nu"])},
            "1": {}, "2": {},
        },
    }


# ---------------------------------------------------------------------------
# streaming writer
# ---------------------------------------------------------------------------
def _map_header(n: int) -> bytes:
    if n < 16:
        return bytes([0x80 | n])
    if n < 2 ** 16:
        return b"\xde" + struct.pack(">H", n)
    return b"\xdf" + struct.pack(">I", n)


def _bin_header(n: int) -> bytes:
    if n < 2 ** 8:
        return b"\xc4" + struct.pack(">B", n)
    if n < 2 ** 16:
        return b"\xc5" + struct.pack(">H", n)
    return b"\xc6" + struct.pack(">I", n)


def _ext_header(code: int, n: int) -> bytes:
    fix = {1: 0xd4, 2: 0xd5, 4: 0xd6, 8: 0xd7, 16: 0xd8}
    if n in fix:
        return bytes([fix[n], code])
    if n < 2 ** 8:
        return b"\xc7" + struct.pack(">Bb", n, code)
    if n < 2 ** 16:
        return b"\xc8" + struct.pack(">Hb", n, code)
    return b"\xc9" + struct.pack(">Ib", n, code)


def _leaf_info(x):
    if isinstance(x, torch.Tensor):
        return tuple(x.shape), _TORCH_TO_NAME[x.dtype], x.numel() * x.element_size()
    return tuple(x.shape), x.dtype.name, x.size * x.dtype.itemsize


def _raw_bytes(x, lo: int, hi: int) -> bytes:
    """C-order bytes of elements [lo, hi) of the flattened leaf (device tensors are copied piecewise)."""
    if isinstance(x, torch.Tensor):
        part = x.reshape(-1)[lo:hi].contiguous().cpu()
        if part.dtype == torch.bfloat16:
            part = part.view(torch.int16)
        return part.numpy().tobytes()
    return np.ascontiguousarray(x).reshape(-1)[lo:hi].tobytes()


def _write_array(f: BinaryIO, shape, dtype_name: str, data: bytes, code: int = EXT_NDARRAY) -> None:
    head = msgpack.packb(list(shape), use_bin_type=True)          # the shape tuple packs as a msgpack array
    inner = b"\x93" + head + msgpack.packb(dtype_name, use_bin_type=True) + _bin_header(len(data))
    f.write(_ext_header(code, len(inner) + len(data)))
    f.write(inner)
    f.write(data)


def _write_leaf(f: BinaryIO, x) -> None:
    shape, name, nbytes = _leaf_info(x)
    itemsize = nbytes // max(1, int(np.prod(shape, dtype=np.int64))) if nbytes else 1
    if nbytes <= MAX_CHUNK_BYTES:
        n = int(np.prod(shape, dtype=np.int64))
        _write_array(f, shape, name, _raw_bytes(x, 0, n))
        return
    n = int(np.prod(shape, dtype=np.int64))
    per = max(1, MAX_CHUNK_BYTES // itemsize)
    nchunks = (n + per - 1) // per
    f.write(_map_header(3))
    f.write(msgpack.packb(_CHUNK_KEY) + b"\xc3")
    f.write(msgpack.packb("shape") + msgpack.packb({str(i): int(d) for i, d in enumerate(shape)}))
    f.write(msgpack.packb("chunks") + _map_header(nchunks))
    for c in range(nchunks):
        lo, hi = c * per, min(n, (c + 1) * per)
        f.write(msgpack.packb(str(c)))
        _write_array(f, (hi - lo,), name, _raw_bytes(x, lo, hi))


def write_tree(f: BinaryIO, tree) -> None:
    if isinstance(tree, dict):
        f.write(_map_header(len(tree)))
        for k, v in tree.items():
            f.write(msgpack.packb(str(k)))
            write_tree(f, v)
    elif isinstance(tree, (torch.Tensor, np.ndarray)):
        _write_leaf(f, tree)
    elif isinstance(tree, np.generic):
        _write_array(f, (), tree.dtype.name, tree.tobytes(), EXT_NPSCALAR)
    else:
        f.write(msgpack.packb(tree, use_bin_type=True))   # python int / float / bool / None / str


# ---------------------------------------------------------------------------
# streaming reader
# ---------------------------------------------------------------------------
class _Reader:
    def __init__(self, f: BinaryIO):
        self.f = f

    def take(self, n: int) -> bytes:
        b = self.f.read(n)
        if len(b) != n:
            raise ValueError("truncated msgpack stream")
        return b

    def u(self, fmt: str):
        return struct.unpack(fmt, self.take(struct.calcsize(fmt)))[0]

    def value(self, on_array: Callable):
        """Decode one value; arrays are handed to ``on_array(shape, dtype_name, reader, nbytes)``."""
        t = self.take(1)[0]
        if t <= 0x7f:
            return t
        if 0x80 <= t <= 0x8f:
            return self.map(t & 0x0f, on_array)
        if 0x90 <= t <= 0x9f:
            return [self.value(on_array) for _ in range(t & 0x0f)]
        if 0xa0 <= t <= 0xbf:
            return self.take(t & 0x1f).decode()
        if t >= 0xe0:
            return t - 256
        if t == 0xc0:
            return None
        if t in (0xc2, 0xc3):
            return t == 0xc3
        if t in (0xc4, 0xc5, 0xc6):
            return self.take(self.u({0xc4: ">B", 0xc5: ">H", 0xc6: ">I"}[t]))
        if t in (0xc7, 0xc8, 0xc9):
            n = self.u({0xc7: ">B", 0xc8: ">H", 0xc9: ">I"}[t])
            return self.ext(self.u(">b"), n, on_array)
        if t in (0xd4, 0xd5, 0xd6, 0xd7, 0xd8):
            return self.ext(self.u(">b"), 1 << (t - 0xd4), on_array)
        if t == 0xca:
            return self.u(">f")
        if t == 0xcb:
            return self.u(">d")
        if t in (0xcc, 0xcd, 0xce, 0xcf):
            return self.u({0xcc: ">B", 0xcd: ">H", 0xce: ">I", 0xcf: ">Q"}[t])
        if t in (0xd0, 0xd1, 0xd2, 0xd3):
            return self.u({0xd0: ">b", 0xd1: ">h", 0xd2: ">i", 0xd3: ">q"}[t])
        if t in (0xd9, 0xda, 0xdb):
            return self.take(self.u({0xd9: ">B", 0xda: ">H", 0xdb: ">I"}[t])).decode()
        if t in (0xdc, 0xdd):
            return [self.value(on_array) for _ in range(self.u(">H" if t == 0xdc else ">I"))]
        if t in (0xde, 0xdf):
            return self.map(self.u(">H" if t == 0xde else ">I"), on_array)
        raise ValueError(f"unsupported msgpack type byte 0x{t:02x}")

    def map(self, n: int, on_array):
        out = {}
        for _ in range(n):
            k = self.value(on_array)
            out[k if isinstance(k, str) else str(k)] = self.value(on_array)
        return out

    def ext(self, code: int, n: int, on_array):
        if code not in (EXT_NDARRAY, EXT_NPSCALAR):
            raise ValueError(f"unsupported msgpack ext type {code}")
        start = self.f.tell()
        if self.take(1)[0] != 0x93:
            raise ValueError("ndarray payload is not a 3-tuple")
        shape = self.value(on_array)
        name = self.value(on_array)
        t = self.take(1)[0]
        nbytes = self.u({0xc4: ">B", 0xc5: ">H", 0xc6: ">I"}[t])
        out = on_array(tuple(shape), name, self, nbytes)
        if self.f.tell() != start + n:
            raise ValueError("ndarray payload length mismatch")
        return out


def _np_dtype(name: str):
    return np.dtype("int16") if name == "bfloat16" else np.dtype(name)


def _to_numpy(shape, name, reader: _Reader, nbytes: int):
    a = np.frombuffer(reader.take(nbytes), dtype=_np_dtype(name)).reshape(shape)
    if name == "bfloat16":
        return torch.from_numpy(a.copy()).view(torch.bfloat16)
    return a


def _unchunk(tree):
    if isinstance(tree, dict):
        if tree.get(_CHUNK_KEY) is True:
            shape = tuple(tree["shape"][str(i)] for i in range(len(tree["shape"])))
            parts = [tree["chunks"][str(i)] for i in range(len(tree["chunks"]))]
            if isinstance(parts[0], torch.Tensor):
                return torch.cat(parts).reshape(shape)
            return np.concatenate(parts).reshape(shape)
        return {k: _unchunk(v) for k, v in tree.items()}
    return tree


def read_tree(f: BinaryIO):
    """Whole file -> nested dict of numpy arrays (bfloat16: torch tensors).  For small states / tests."""
    return _unchunk(_Reader(f).value(_to_numpy))


def restore_into(f: BinaryIO, targets: dict) -> dict:
    """Stream a checkpoint into preallocated tensors: ``targets`` maps a state-dict path ("params/blocks_0/...",
    "opt_state/0/mu/...") to the torch tensor to fill (any device).  Returns the non-array leaves ({"step": ...})
    and raises on missing / unexpected / mis-shaped leaves.

    Two passes over the file: the first walks the whole tree checking keys, shapes, dtypes and payload lengths against
    the file size while SEEKING over the payloads (no bytes read, nothing written); only when it succeeds does the
    second pass copy into ``targets``.  A truncated file or a tree that does not match the template therefore leaves
    the template untouched -- like the reference's functional ``serialization.from_bytes`` (trainers/utils.py:53-58),
    whose template is never modified by a failed load."""
    start = f.tell()
    f.seek(0, 2)
    end = f.tell()
    f.seek(start)
    _restore_pass(f, targets, dry=True, end=end)
    f.seek(start)
    return _restore_pass(f, targets, dry=False, end=end)


def _restore_pass(f: BinaryIO, targets: dict, dry: bool, end: int) -> dict:
    seen: set = set()
    scalars: dict = {}
    path: list = []

    def fill(dst: torch.Tensor, lo: int, name: str, reader: _Reader, nbytes: int):
        want = _TORCH_TO_NAME[dst.dtype]
        if name != want:
            raise ValueError(f"dtype mismatch at {'/'.join(path)}: checkpoint {name}, state {want}")
        n = nbytes // dst.element_size()
        flat = dst.reshape(-1)
        if lo + n > flat.numel() or nbytes % dst.element_size():
            raise ValueError(f"size mismatch at {'/'.join(path)}")
        if reader.f.tell() + nbytes > end:
            raise ValueError("truncated msgpack stream")
        if dry:
            reader.f.seek(nbytes, 1)
            return n
        step = max(1, (256 << 20) // dst.element_size())
        for o in range(0, n, step):
            m = min(step, n - o)
            host = np.frombuffer(reader.take(m * dst.element_size()), dtype=_np_dtype(name))
            t = torch.from_numpy(host.copy())
            if name == "bfloat16":
                t = t.view(torch.bfloat16)
            flat[lo + o:lo + o + m].copy_(t)
        return n

    def walk(reader: _Reader):
        t = reader.take(1)[0]
        reader.f.seek(-1, 1)
        is_map = 0x80 <= t <= 0x8f or t in (0xde, 0xdf)
        if not is_map:
            key = "/".join(path)

            def on_array(shape, name, rd, nbytes):
                if key not in targets:
                    if key.endswith("/count") or key == "step":
                        return _to_numpy(shape, name, rd, nbytes)
                    raise ValueError(f"unexpected leaf in checkpoint: {key}")
                dst = targets[key]
                if tuple(dst.shape) != tuple(shape):
                    raise ValueError(f"shape mismatch at {key}: checkpoint {tuple(shape)}, state {tuple(dst.shape)}")
                fill(dst, 0, name, rd, nbytes)
                seen.add(key)
                return None
            v = reader.value(on_array)
            if v is not None:
                scalars[key] = v
            return
        reader.take(1)
        n = (t & 0x0f) if t <= 0x8f else reader.u(">H" if t == 0xde else ">I")
        keys_pos = reader.f.tell()
        # peek: is this a chunked array?
        first = reader.value(_to_numpy) if n else None
        if first == _CHUNK_KEY:
            key = "/".join(path)
            if key not in targets:
                raise ValueError(f"unexpected leaf in checkpoint: {key}")
            dst = targets[key]
            reader.value(_to_numpy)                       # True
            shape = None
            for _ in range(n - 1):
                k = reader.value(_to_numpy)
                if k == "shape":
                    sd = reader.value(_to_numpy)
                    shape = tuple(sd[str(i)] for i in range(len(sd)))
                    if tuple(dst.shape) != shape:
                        raise ValueError(f"shape mismatch at {key}: checkpoint {shape}, state {tuple(dst.shape)}")
                elif k == "chunks":
                    t2 = reader.take(1)[0]
                    nc = (t2 & 0x0f) if t2 <= 0x8f else reader.u(">H" if t2 == 0xde else ">I")
                    off = [0]
                    for _c in range(nc):
                        reader.value(_to_numpy)            # chunk index (ascending, as flax writes them)
                        reader.value(lambda s, nm, rd, nb: off.__setitem__(0, off[0] + fill(dst, off[0], nm, rd, nb)))
                    if off[0] != dst.numel():
                        raise ValueError(f"size mismatch at {key}")
                else:
                    raise ValueError(f"malformed chunked array at {key}")
            seen.add(key)
            return
        reader.f.seek(keys_pos)
        for _ in range(n):
            k = reader.value(_to_numpy)
            path.append(str(k))
            walk(reader)
            path.pop()

    walk(_Reader(f))
    missing = set(targets) - seen
    if missing:
        raise ValueError(f"checkpoint lacks {len(missing)} leaves, e.g. {sorted(missing)[:3]}")
    return scalars


# ---------------------------------------------------------------------------
# the reference's checkpoint API (trainers/utils.py)
# ---------------------------------------------------------------------------
def save_json(path: Path, payload: dict) -> None:
    path.parent.mkdir(parents=True, exist_ok=True)
    with path.open("w", encoding="utf-8") as f:
        json.dump(payload, f, indent=2, sort_keys=True)


def load_json(path: Path) -> dict:
    if not path.exists():
        return {}
    with path.open("r", encoding="utf-8") as f:
        return json.load(f)


def save_checkpoint(path: Path, state) -> None:
    """``trainers/utils.py:45-50``."""
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    tmp = path.with_suffix(path.suffix + ".tmp")
    with tmp.open("wb") as f:
        write_tree(f, state_dict(state))
    tmp.replace(path)


def load_checkpoint(path: Path, state_template):
    """``trainers/utils.py:53-58``: fills ``state_template`` in place (this backend's TrainState is mutable) and
    returns it; raises ``ValueError`` when the file does not match the template's tree, shapes or dtypes."""
    targets = {}
    for k, v in state_template.params.items():
        targets[f"params/{k}"] = v
    for k, v in state_template.opt_state["mu"].items():
        targets[f"opt_state/0/mu/{k}"] = v
    for k, v in state_template.opt_state["nu"].items():
        targets[f"opt_state/0/nu/{k}"] = v
    with Path(path).open("rb") as f:
        scalars = restore_into(f, targets)
    step = scalars.get("step")
    if step is None:
        raise ValueError("checkpoint has no 'step'")
    state_template.step = int(np.asarray(step))
    state_template.refresh_work()
    return state_template


def save_unwrapped_checkpoint(path: Path, params: dict) -> None:
    """Parameters only (``trainers/utils.py:561-573``): ``serialization.to_bytes(params)`` of the nested tree."""
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    with path.open("wb") as f:
        write_tree(f, nest(params) if all(not isinstance(v, dict) for v in params.values()) else params)


def load_unwrapped_checkpoint(path: Path, into: dict | None = None) -> dict:
    """``trainers/utils.py:576-588``.  With ``into`` (flat ``{"blocks_0/...": tensor}``) the leaves are streamed
    into those tensors -- e.g. a model's parameters for the 1-NFE decoder; without it the flat tree is returned."""
    with Path(path).open("rb") as f:
        if into is not None:
            restore_into(f, dict(into))
            return into
        return flatten(read_tree(f))


def get_checkpoint_metadata_path(checkpoint_path: Path) -> Path:
    return Path(checkpoint_path).with_suffix(".json")


def get_checkpoint_step(checkpoint_path: Path) -> int:
    m = re.search(r"step_(\d+)\.msgpack", Path(checkpoint_path).name)
    if not m:
        raise ValueError(f"Could not extract step number from: {checkpoint_path}")
    return int(m.group(1))


def get_git_commit_hash(cwd: Path | None = None) -> tuple[str | None, str | None]:
    try:
        full = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True, cwd=cwd, timeout=5)
        if full.returncode == 0:
            h = full.stdout.strip()
            return h, h[:7]
    except Exception:
        pass
    return None, None


def compute_config_hash(config: dict) -> str:
    return hashlib.sha256(json.dumps(config, sort_keys=True).encode()).hexdigest()


def _jsonable(d):
    return json.loads(json.dumps(d, default=str))


def save_checkpoint_metadata(checkpoint_path: Path, step: int, state, config=None) -> None:
    """Sidecar ``step_%05d.json`` with the reference's keys (``trainers/utils.py:111-177``); the jax/flax version
    fields are null and ``backend`` names this implementation."""
    commit, short = get_git_commit_hash()
    shapes = {k: list(v.shape) for k, v in state.params.items()}
    meta = {
        "step": step,
        "timestamp": datetime.now().isoformat(),
        "config_hash": compute_config_hash(_jsonable(config.to_dict())) if config is not None else None,
        "git_commit": commit, "git_commit_short": short,
        "system_info": {"platform": f"{platform.system()} {platform.release()}",
                        "python_version": sys.version.split()[0], "jax_version": None, "flax_version": None,
                        "backend": f"meanflow_audio_codec_amd (torch {torch.__version__})"},
        "checkpoint_size_bytes": Path(checkpoint_path).stat().st_size if Path(checkpoint_path).exists() else 0,
        "model_info": {"param_count": int(sum(v.numel() for v in state.params.values())), "param_shapes": shapes},
    }
    save_json(get_checkpoint_metadata_path(checkpoint_path), meta)


def load_checkpoint_metadata(checkpoint_path: Path) -> dict | None:
    p = get_checkpoint_metadata_path(checkpoint_path)
    return load_json(p) if p.exists() else None


def save_checkpoint_with_metadata(checkpoint_path: Path, state, step: int, config=None) -> None:
    save_checkpoint(checkpoint_path, state)
    save_checkpoint_metadata(checkpoint_path, step, state, config)


def validate_checkpoint(checkpoint_path: Path) -> tuple[bool, str | None]:
    p = Path(checkpoint_path)
    if not p.exists():
        return False, f"Checkpoint file does not exist: {p}"
    if not p.is_file():
        return False, f"Checkpoint path is not a file: {p}"
    size = p.stat().st_size
    if size == 0:
        return False, f"Checkpoint file is empty: {p}"
    if size < 100:
        return False, f"Checkpoint file is suspiciously small ({size} bytes): {p}"
    return True, None


def _checkpoints(workdir: Path) -> list[tuple[int, Path]]:
    d = Path(workdir) / "checkpoints"
    out = []
    if d.exists():
        for p in d.glob("step_*.msgpack"):
            try:
                out.append((get_checkpoint_step(p), p))
            except ValueError:
                continue
    return sorted(out)


def find_latest_checkpoint(workdir: Path) -> Path | None:
    c = _checkpoints(workdir)
    return c[-1][1] if c else None


def list_checkpoints(workdir: Path) -> list[dict]:
    return [{"path": str(p), "exists": True, "step": s, "size_bytes": p.stat().st_size,
             "metadata": load_checkpoint_metadata(p)} for s, p in _checkpoints(workdir)]


def load_checkpoint_and_resume(workdir: Path, state_template, config=None):
    """Newest checkpoint that validates AND loads into the template (corrupted / mismatching ones are skipped, as
    ``find_valid_checkpoint`` does); returns ``(state, starting_step)``; ``FileNotFoundError`` when none does."""
    last_err = None
    for step, p in reversed(_checkpoints(workdir)):
        ok, err = validate_checkpoint(p)
        if not ok:
            last_err = err
            continue
        try:
            state = load_checkpoint(p, state_template)
        except Exception as e:   # noqa: BLE001 -- a corrupted file must not stop the search
            last_err = str(e)
            continue
        meta = load_checkpoint_metadata(p)
        if config is not None and meta and meta.get("config_hash"):
            if meta["config_hash"] != compute_config_hash(_jsonable(config.to_dict())):
                print("Checkpoint compatibility warnings:\n  - config hash differs from the checkpoint's")
        return state, step
    raise FileNotFoundError(f"No valid checkpoint found in {Path(workdir) / 'checkpoints'}"
                            + (f" ({last_err})" if last_err else ""))


def cleanup_old_checkpoints(workdir: Path, max_checkpoints_to_keep: int, keep_final: bool = True,
                            final_step: int | None = None) -> list[Path]:
    c = _checkpoints(workdir)
    if len(c) <= max_checkpoints_to_keep:
        return []
    keep = {p for _, p in c[-max_checkpoints_to_keep:]} if max_checkpoints_to_keep > 0 else set()
    if keep_final and final_step is not None:
        keep |= {p for s, p in c if s == final_step}
    removed = []
    for _, p in c:
        if p in keep:
            continue
        try:
            p.unlink()
            removed.append(p)
            m = get_checkpoint_metadata_path(p)
            if m.exists():
                m.unlink()
        except OSError:
            pass
    return removed


# ---------------------------------------------------------------------------
# logs / summaries (trainers/utils.py:473-509, 1034-1094)
# ---------------------------------------------------------------------------
class LogWriter:
    """One JSON object per line, appended and flushed per step."""

    def __init__(self, log_path: Path):
        self.log_path = Path(log_path)
        self.log_path.parent.mkdir(parents=True, exist_ok=True)
        self.file = self.log_path.open("a", encoding="utf-8")

    def write_step(self, step: int, metrics: dict) -> None:
        self.file.write(json.dumps({"step": step, **metrics}, sort_keys=True) + "\n")
        self.file.flush()

    def close(self) -> None:
        if self.file:
            self.file.close()
            self.file = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def read_log(log_path: Path) -> Iterator[dict]:
    p = Path(log_path)
    if not p.exists():
        return
    with p.open("r", encoding="utf-8") as f:
        for line in f:
            line = line.strip()
            if line:
                try:
                    yield json.loads(line)
                except json.JSONDecodeError:
                    continue


def generate_training_summary(log_path: Path) -> dict:
    rows = [r for r in read_log(log_path) if "loss" in r]
    if not rows:
        return {"error": "No metrics found in log file"}
    losses = [float(r["loss"]) for r in rows]
    best = min(rows, key=lambda r: r["loss"])
    out: dict[str, Any] = {
        "best_loss": {"value": best["loss"], "step": best.get("step", -1)},
        "final_metrics": rows[-1],
        "total_steps": max(r.get("step", -1) for r in rows),
        "logged_steps": len(rows),
        "loss_statistics": {"mean": float(np.mean(losses)), "std": float(np.std(losses)),
                            "min": float(np.min(losses)), "max": float(np.max(losses)), "count": len(losses)},
    }
    avg_rows = [r for r in rows if "loss_avg" in r]
    if avg_rows:
        b = min(avg_rows, key=lambda r: r["loss_avg"])
        out["best_loss_avg"] = {"value": b["loss_avg"], "step": b.get("step", -1)}
    if len(losses) > 10:
        early, late = sum(losses[:10]) / 10, sum(losses[-10:]) / 10
        out["convergence"] = {"early_avg_loss": early, "late_avg_loss": late, "improvement": early - late,
                              "improvement_percent": ((early - late) / early * 100) if early > 0 else 0.0}
    return out
