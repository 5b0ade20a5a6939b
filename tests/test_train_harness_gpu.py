"""GPU: the reference's workdir layout, checkpoint + resume through train_flow and the CLI."""
import json

import pytest
import torch

pytestmark = pytest.mark.gpu


def _config(workdir, **kw):
    from meanflow_audio_codec_amd.configs import TrainFlowConfig
    base = dict(batch_size=8, n_steps=4, sample_every=2, sample_seed=3, sample_steps=2, base_lr=1e-3, weight_decay=1e-4,
                seed=0, noise_dimension=64, condition_dimension=16, latent_dimension=8, num_blocks=1, dataset="mnist",
                architecture="mlp", use_improved_mean_flow=True, loss_strategy="improved_mean_flow", workdir=workdir)
    base.update(kw)
    return TrainFlowConfig(**base)


def test_workdir_layout_checkpoint_and_resume(tmp_path):
    from meanflow_audio_codec_amd.trainers import checkpoint as ck
    from meanflow_audio_codec_amd.trainers.train import synthetic_iterator, train_flow
    wd = tmp_path / "run"
    cfg = _config(wd, checkpoint_step=2)
    state, _ = train_flow(cfg, synthetic_iterator(cfg))
    assert state.step == 4
    for rel in ("config.json", "metadata.json", "summary.json", "logs/train_log.jsonl", "checkpoints/step_00002.msgpack",
                "checkpoints/step_00002.json", "samples/step_0000.npy", "samples/step_0002.npy", "samples/step_0004.npy"):
        assert (wd / rel).exists(), rel
    rows = [json.loads(l) for l in (wd / "logs" / "train_log.jsonl").read_text().splitlines()]
    assert [r["step"] for r in rows] == [0, 1, 2, 3] and set(rows[0]) >= {"loss", "loss_avg", "lr", "step_time"}
    assert json.loads((wd / "config.json").read_text())["base"]["batch_size"] == 8
    summ = json.loads((wd / "summary.json").read_text())
    assert summ["metrics"]["logged_steps"] == 4 and summ["profiling"]["param_count"] > 0
    meta = ck.load_checkpoint_metadata(wd / "checkpoints" / "step_00002.msgpack")
    assert meta["step"] == 2 and meta["model_info"]["param_count"] == summ["profiling"]["param_count"]

    # the checkpoint holds the step-2 state: params differ from the final ones, moments are non-zero
    with (wd / "checkpoints" / "step_00002.msgpack").open("rb") as f:
        tree = ck.read_tree(f)
    assert tree["step"] == 2 and int(tree["opt_state"]["0"]["count"]) == 2
    flat = ck.flatten(tree["params"])
    assert set(flat) == set(state.params)
    k = next(k for k in flat if k.endswith("kernel"))
    assert not torch.equal(torch.from_numpy(flat[k].copy()).cuda(), state.params[k])
    assert abs(ck.flatten(tree["opt_state"]["0"]["mu"])[k]).max() > 0

    # resume: continues at step 2 with exactly the checkpointed parameters and a config diff on record
    cfg2 = _config(wd, checkpoint_step=None, n_steps=3, base_lr=5e-4)
    seen = {}
    from meanflow_audio_codec_amd.trainers import training_steps as ts
    orig = ts.train_step

    def spy(state, key, x, strategy, **kw):
        if "first" not in seen:
            seen["first"] = (state.step, {n: v.clone() for n, v in state.params.items()}, key.counter)
        return orig(state, key, x, strategy, **kw)
    import meanflow_audio_codec_amd.trainers.train as tr
    tr.train_step = spy
    try:
        state2, _ = train_flow(cfg2, synthetic_iterator(cfg2), resume=True)
    finally:
        tr.train_step = orig
    step0, params0, counter0 = seen["first"]
    assert step0 == 2 and counter0 == 2 and state2.step == 3
    for n, v in params0.items():
        assert torch.equal(v.cpu(), torch.from_numpy(flat[n].copy())), n
    diff = json.loads((wd / "config_diff.json").read_text())
    assert diff["changed"]["base_lr"] == {"old": 1e-3, "new": 5e-4}
    assert (wd / "checkpoints" / "step_00003.msgpack").exists()
    rows = [json.loads(l) for l in (wd / "logs" / "train_log.jsonl").read_text().splitlines()]
    assert [r["step"] for r in rows] == [0, 1, 2, 3, 2]          # the log is appended to, as in the reference

    # resume with nothing to resume from starts from scratch; max_checkpoints_to_keep prunes
    cfg3 = _config(tmp_path / "fresh", n_steps=2, max_checkpoints_to_keep=1, checkpoint_step=1)
    state3, _ = train_flow(cfg3, synthetic_iterator(cfg3), resume=True)
    assert state3.step == 2
    assert sorted(p.name for p in (tmp_path / "fresh" / "checkpoints").glob("*.msgpack")) == ["step_00001.msgpack"]


def test_requires_data_source_and_even_condition_dimension(tmp_path):
    from meanflow_audio_codec_amd.trainers.train import synthetic_iterator, train_flow
    cfg = _config(tmp_path / "a")
    with pytest.raises(ValueError, match="data_dir"):
        train_flow(cfg)
    with pytest.raises(ValueError, match="even"):      # already rejected by the config's own validation
        _config(tmp_path / "b", condition_dimension=15)
    with pytest.raises(FileNotFoundError):             # the MNIST IDX files are looked up in data_dir
        train_flow(_config(tmp_path / "c", data_dir=str(tmp_path)))


def test_cli_runs_a_config_file(tmp_path):
    from meanflow_audio_codec_amd import train_cli
    cfgp = tmp_path / "c.json"
    cfgp.write_text(json.dumps(dict(batch_size=4, n_steps=2, sample_every=5, sample_seed=1, sample_steps=1, base_lr=1e-3,
                                    weight_decay=1e-4, seed=1, noise_dimension=32, condition_dimension=16,
                                    latent_dimension=8, num_blocks=1, dataset="mnist", architecture="mlp",
                                    use_improved_mean_flow=False)))
    with pytest.raises(ValueError, match="data_dir"):      # as the reference: no data_dir, no training
        train_cli.main(["--config", str(cfgp), "--workdir", str(tmp_path / "w0")])
    assert train_cli.main(["--config", str(cfgp), "--workdir", str(tmp_path / "w"), "--synthetic"]) == 0
    assert (tmp_path / "w" / "checkpoints" / "step_00002.msgpack").exists()
    assert train_cli.main(["--config", str(cfgp), "--workdir", str(tmp_path / "w"), "--resume", "--steps", "2",
                           "--synthetic"]) == 0
