"""Checkpoint format (flax msgpack layout) and checkpoint management -- CPU only."""
import io
import json

import numpy as np
import pytest
import torch

from meanflow_audio_codec_amd.trainers import checkpoint as ck
from oracle import flax_msgpack as fm


class _State:
    """The attributes of TrainState the checkpoint code touches (the real one needs the GPU library)."""

    def __init__(self, seed=0, step=7, n=5):
        g = torch.Generator().manual_seed(seed)
        self.params = {"blocks_0/input_proj1/kernel": torch.randn(n, 3, generator=g),
                       "blocks_0/input_proj1/bias": torch.randn(3, generator=g),
                       "blocks_0/conv_block/layer_scale_gamma": torch.randn(4, generator=g),
                       "latent_proj/kernel": torch.randn(2, n, generator=g)}
        self.opt_state = {"mu": {k: torch.randn(v.shape, generator=g) for k, v in self.params.items()},
                          "nu": {k: torch.rand(v.shape, generator=g) for k, v in self.params.items()}}
        self.step = step
        self.refreshed = 0

    def refresh_work(self):
        self.refreshed += 1


def _np_tree(t):
    if isinstance(t, dict):
        return {k: _np_tree(v) for k, v in t.items()}
    return t.numpy() if isinstance(t, torch.Tensor) else t


@pytest.mark.parametrize("chunk", [2 ** 30, 24])
def test_streaming_writer_matches_flax_layout_byte_for_byte(monkeypatch, chunk):
    """chunk=24 bytes forces the '__msgpack_chunked_array__' branch on every multi-element leaf."""
    monkeypatch.setattr(ck, "MAX_CHUNK_BYTES", chunk)
    st = _State()
    buf = io.BytesIO()
    ck.write_tree(buf, ck.state_dict(st))
    want = fm.msgpack_serialize(_np_tree(ck.state_dict(st)), max_chunk_bytes=chunk)
    assert buf.getvalue() == want
    tree = fm.msgpack_restore(buf.getvalue())
    assert set(tree) == {"step", "params", "opt_state"} and tree["step"] == 7
    assert set(tree["opt_state"]) == {"0", "1", "2"} and tree["opt_state"]["1"] == {} and tree["opt_state"]["2"] == {}
    assert tree["opt_state"]["0"]["count"].dtype == np.int32 and int(tree["opt_state"]["0"]["count"]) == 7
    np.testing.assert_array_equal(tree["params"]["blocks_0"]["input_proj1"]["kernel"],
                                  st.params["blocks_0/input_proj1/kernel"].numpy())


@pytest.mark.parametrize("chunk", [2 ** 30, 24])
def test_reader_round_trip_and_streaming_restore(tmp_path, monkeypatch, chunk):
    monkeypatch.setattr(ck, "MAX_CHUNK_BYTES", chunk)
    st = _State(seed=1, step=11)
    p = tmp_path / "checkpoints" / "step_00011.msgpack"
    ck.save_checkpoint(p, st)
    with p.open("rb") as f:
        tree = ck.read_tree(f)
    flat = ck.flatten(tree["params"])
    for k, v in st.params.items():
        np.testing.assert_array_equal(flat[k], v.numpy())
    # a file produced by the flax-layout oracle loads as well
    (tmp_path / "o.msgpack").write_bytes(fm.msgpack_serialize(_np_tree(ck.state_dict(st)), max_chunk_bytes=chunk))
    for path in (p, tmp_path / "o.msgpack"):
        tmpl = _State(seed=99, step=0)
        out = ck.load_checkpoint(path, tmpl)
        assert out is tmpl and tmpl.step == 11 and tmpl.refreshed == 1
        for k in st.params:
            assert torch.equal(tmpl.params[k], st.params[k])
            assert torch.equal(tmpl.opt_state["mu"][k], st.opt_state["mu"][k])
            assert torch.equal(tmpl.opt_state["nu"][k], st.opt_state["nu"][k])


def test_bfloat16_leaves(tmp_path):
    t = {"w": torch.randn(3, 5).bfloat16()}
    buf = io.BytesIO()
    ck.write_tree(buf, t)
    buf.seek(0)
    back = ck.read_tree(buf)
    assert back["w"].dtype == torch.bfloat16 and torch.equal(back["w"], t["w"])
    shape, name, _ = __import__("msgpack").unpackb(__import__("msgpack").unpackb(buf.getvalue(), raw=False)["w"].data, raw=False)
    assert name == "bfloat16" and list(shape) == [3, 5]


def test_mismatches_raise_value_error(tmp_path):
    st = _State()
    p = tmp_path / "step_00007.msgpack"
    ck.save_checkpoint(p, st)
    bad = _State(n=6)                                   # different kernel shape
    with pytest.raises(ValueError, match="shape mismatch"):
        ck.load_checkpoint(p, bad)
    extra = _State()
    extra.params["blocks_1/x"] = torch.zeros(2)
    extra.opt_state["mu"]["blocks_1/x"] = torch.zeros(2)
    extra.opt_state["nu"]["blocks_1/x"] = torch.zeros(2)
    with pytest.raises(ValueError, match="lacks"):
        ck.load_checkpoint(p, extra)
    fewer = _State()
    for d in (fewer.params, fewer.opt_state["mu"], fewer.opt_state["nu"]):
        d.pop("latent_proj/kernel")
    with pytest.raises(ValueError, match="unexpected leaf"):
        ck.load_checkpoint(p, fewer)
    p.write_bytes(p.read_bytes()[:200])
    with pytest.raises(ValueError):
        ck.load_checkpoint(p, _State())


def _snapshot(st):
    return ({k: v.clone() for k, v in st.params.items()}, {k: v.clone() for k, v in st.opt_state["mu"].items()},
            {k: v.clone() for k, v in st.opt_state["nu"].items()}, st.step, st.refreshed)


def _same(st, snap):
    return (all(torch.equal(st.params[k], snap[0][k]) for k in snap[0]) and
            all(torch.equal(st.opt_state["mu"][k], snap[1][k]) for k in snap[1]) and
            all(torch.equal(st.opt_state["nu"][k], snap[2][k]) for k in snap[2]) and
            st.step == snap[3] and st.refreshed == snap[4])


@pytest.mark.parametrize("chunk", [2 ** 30, 24])
def test_failed_load_leaves_the_template_untouched(tmp_path, monkeypatch, chunk):
    """a failed load must not leave the template partly overwritten (train_flow would then 'start from
    scratch' on a hybrid state).  The loader validates the whole file against the template before its first write."""
    monkeypatch.setattr(ck, "MAX_CHUNK_BYTES", chunk, raising=False)
    src = _State(seed=3, step=30)
    p = tmp_path / "checkpoints" / "step_00030.msgpack"
    ck.save_checkpoint(p, src)
    raw = p.read_bytes()
    # (a) truncated at several depths: some leaves complete in the file, later ones cut
    for cut in (len(raw) - 1, len(raw) - 40, len(raw) // 2, len(raw) // 3):
        p.write_bytes(raw[:cut])
        tmpl = _State(seed=5, step=0)
        snap = _snapshot(tmpl)
        with pytest.raises(ValueError):
            ck.load_checkpoint(p, tmpl)
        assert _same(tmpl, snap), cut
    # (b) a mismatching LATE leaf (the tree is written in sorted key order: params < opt_state is not the file order,
    # so make the mismatch the last leaf of the params subtree)
    p.write_bytes(raw)
    tmpl = _State(seed=5, step=0)
    tmpl.params["latent_proj/kernel"] = torch.zeros(3, 5)
    snap = _snapshot(tmpl)
    with pytest.raises(ValueError, match="shape mismatch"):
        ck.load_checkpoint(p, tmpl)
    assert _same(tmpl, snap)
    # (c) dtype mismatch
    tmpl = _State(seed=5, step=0)
    tmpl.opt_state["nu"]["latent_proj/kernel"] = tmpl.opt_state["nu"]["latent_proj/kernel"].bfloat16()
    snap = _snapshot(tmpl)
    with pytest.raises(ValueError, match="dtype mismatch"):
        ck.load_checkpoint(p, tmpl)
    assert _same(tmpl, snap)
    # (d) resume: newest truncated, older one of a DIFFERENT architecture -> nothing loads, template bit-equal to fresh
    other = _State(seed=9, step=20, n=6)
    ck.save_checkpoint(tmp_path / "checkpoints" / "step_00020.msgpack", other)
    p.write_bytes(raw[:len(raw) // 2])
    tmpl = _State(seed=5, step=0)
    snap = _snapshot(tmpl)
    with pytest.raises(FileNotFoundError):
        ck.load_checkpoint_and_resume(tmp_path, tmpl)
    assert _same(tmpl, snap)
    # and the intact file still loads completely
    p.write_bytes(raw)
    tmpl = _State(seed=5, step=0)
    st = ck.load_checkpoint(p, tmpl)
    assert st.step == 30 and st.refreshed == 1
    assert all(torch.equal(st.params[k], src.params[k]) for k in src.params)
    assert all(torch.equal(st.opt_state["nu"][k], src.opt_state["nu"][k]) for k in src.params)


def test_management_resume_and_cleanup(tmp_path):
    class Cfg:
        def to_dict(self):
            return {"batch_size": 4, "workdir": "w"}
    wd = tmp_path
    for s in (10, 20, 30):
        st = _State(seed=s, step=s)
        ck.save_checkpoint_with_metadata(wd / "checkpoints" / f"step_{s:05d}.msgpack", st, s, Cfg())
    assert ck.find_latest_checkpoint(wd).name == "step_00030.msgpack"
    meta = ck.load_checkpoint_metadata(wd / "checkpoints" / "step_00020.msgpack")
    assert meta["step"] == 20 and meta["model_info"]["param_count"] == 15 + 3 + 4 + 10
    assert meta["config_hash"] == ck.compute_config_hash(Cfg().to_dict())
    assert set(meta) >= {"timestamp", "git_commit", "system_info", "checkpoint_size_bytes"}
    # newest checkpoint corrupted -> resume falls back to step 20 (find_valid_checkpoint behaviour)
    (wd / "checkpoints" / "step_00030.msgpack").write_bytes(b"\x00" * 150)
    tmpl = _State(seed=5)
    state, start = ck.load_checkpoint_and_resume(wd, tmpl, Cfg())
    assert start == 20 and state.step == 20
    assert torch.equal(state.params["latent_proj/kernel"], _State(seed=20).params["latent_proj/kernel"])
    removed = ck.cleanup_old_checkpoints(wd, 1, keep_final=True, final_step=10)
    assert [p.name for p in removed] == ["step_00020.msgpack"]
    assert not (wd / "checkpoints" / "step_00020.json").exists() and (wd / "checkpoints" / "step_00010.json").exists()
    with pytest.raises(FileNotFoundError):
        ck.load_checkpoint_and_resume(tmp_path / "nothing", _State())
    assert [c["step"] for c in ck.list_checkpoints(wd)] == [10, 30]
    with pytest.raises(ValueError):
        ck.get_checkpoint_step(wd / "model.bin")


def test_log_writer_and_summary(tmp_path):
    lp = tmp_path / "logs" / "train_log.jsonl"
    with ck.LogWriter(lp) as lw:
        for s in range(25):
            lw.write_step(s, {"loss": 2.0 - 0.05 * s, "loss_avg": 2.0 - 0.04 * s, "lr": 1e-4})
    rows = [json.loads(l) for l in lp.read_text().splitlines()]
    assert rows[3] == {"step": 3, "loss": 1.85, "loss_avg": 1.88, "lr": 1e-4}
    s = ck.generate_training_summary(lp)
    assert s["best_loss"] == {"value": pytest.approx(0.8), "step": 24} and s["logged_steps"] == 25
    assert s["convergence"]["improvement"] > 0 and s["loss_statistics"]["count"] == 25
    assert ck.generate_training_summary(tmp_path / "none.jsonl") == {"error": "No metrics found in log file"}


def test_unwrapped_parameter_checkpoint(tmp_path):
    st = _State(seed=3)
    p = tmp_path / "params.msgpack"
    ck.save_unwrapped_checkpoint(p, st.params)
    assert p.read_bytes() == fm.msgpack_serialize(_np_tree(ck.nest(st.params)))
    flat = ck.load_unwrapped_checkpoint(p)
    assert set(flat) == set(st.params)
    dst = {k: torch.zeros_like(v) for k, v in st.params.items()}
    ck.load_unwrapped_checkpoint(p, into=dst)
    assert all(torch.equal(dst[k], st.params[k]) for k in dst)
