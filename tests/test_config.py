"""CPU: the config mirror loads the reference's experiment matrix identically (golden: the reference's
own ``to_dict()`` for all 74 ``configs/*.json``) and keeps its validation / migration / merge behaviour
(test/test_config.py:25-274 restated)."""
import json

import pytest

from meanflow_audio_codec_amd.configs import (TrainFlowConfig, create_audio_config, create_mnist_config,
                                              diff_configs, load_config_from_json, merge_configs,
                                              migrate_config_v1_to_v2)


@pytest.fixture(scope="module")
def golden(golden_dir):
    return json.loads((golden_dir / "configs_reference.json").read_text())


def test_all_reference_jsons_load_identically(golden, tmp_path):
    assert len(golden) == 74
    for name, rec in golden.items():
        p = tmp_path / name
        p.write_text(json.dumps(rec["raw"]))
        cfg = load_config_from_json(p)
        assert json.loads(json.dumps(cfg.to_dict())) == rec["to_dict"], name
        for k, v in rec["flat"].items():
            assert getattr(cfg, k) == v, (name, k)
        # hierarchical round trip
        assert TrainFlowConfig.from_dict(cfg.to_dict()).to_dict() == cfg.to_dict()


def _valid(**kw):
    d = dict(batch_size=8, n_steps=10, base_lr=1e-3, weight_decay=0.0, seed=0, noise_dimension=16,
             condition_dimension=8, latent_dimension=4, num_blocks=1, sample_every=5, sample_seed=0, sample_steps=2)
    d.update(kw)
    return d


@pytest.mark.parametrize("bad", [dict(batch_size=0), dict(n_steps=-1), dict(base_lr=0), dict(weight_decay=-1),
                                 dict(condition_dimension=7), dict(architecture="resnet"), dict(dataset="cifar"),
                                 dict(tokenization_strategy="stft"), dict(method="gan"), dict(noise_min=1.0),
                                 dict(noise_min=0.5, noise_max=0.4), dict(time_sampling_data_proportion=1.5),
                                 dict(method="improved_mean_flow", use_improved_mean_flow=False),
                                 dict(sample_every=0), dict(checkpoint_step=0), dict(gamma=0)])
def test_validation_errors(bad):
    with pytest.raises(ValueError):
        TrainFlowConfig(**_valid(**bad))


def test_flat_access_is_read_only_and_migration():
    cfg = TrainFlowConfig(**_valid(workdir="out/x", architecture="convnet"))
    assert cfg.batch_size == cfg.base.batch_size == 8 and cfg.model.architecture == "convnet"
    assert str(cfg.workdir) == "out/x"
    with pytest.raises(AttributeError):
        cfg.workdir = "elsewhere"          # the reference's train.py:77 trips over exactly this
    v2 = migrate_config_v1_to_v2(_valid(gamma=0.5))
    assert v2["config_version"] == "2.0" and v2["method"]["gamma"] == 0.5 and v2["base"]["batch_size"] == 8
    with pytest.raises(TypeError):
        TrainFlowConfig(**_valid(bogus=1))


def test_merge_diff_factories():
    a = create_mnist_config()
    b = merge_configs(a, {"batch_size": 64, "method": {"gamma": 0.25}})
    assert b.batch_size == 64 and b.gamma == 0.25 and a.batch_size == 128
    d = diff_configs(a, b)
    assert d["changed"]["batch_size"] == {"old": 128, "new": 64} and "gamma" in d["added"]
    au = create_audio_config()
    assert au.tokenization_config == {"window_size": 512, "hop_size": 256} and au.noise_dimension == 196608
    assert "batch_size" in au.get_schema()["base"] and "TrainFlowConfig" in au.get_documentation()


def test_create_loss_strategy_defaults(golden, tmp_path):
    """trainers/train.py:52-153: no JSON sets loss_strategy, so the fallback decides (defect 3)."""
    from meanflow_audio_codec_amd.trainers import (FlowMatchingLoss, ImprovedMeanFlowLoss, MeanFlowLoss,
                                                   MeanFlowTimeSampling, create_loss_strategy)
    seen = set()
    for name, rec in golden.items():
        cfg = TrainFlowConfig.from_dict(rec["raw"])
        s = create_loss_strategy(cfg)
        want = ImprovedMeanFlowLoss if cfg.use_improved_mean_flow else FlowMatchingLoss
        assert type(s) is want, name
        assert s.noise_schedule.noise_min == 0.001 and s.noise_schedule.noise_max == 0.999
        assert s.use_weighted_loss is True
        if isinstance(s, ImprovedMeanFlowLoss):
            assert isinstance(s.time_sampling, MeanFlowTimeSampling)
            assert (s.time_sampling.mean, s.time_sampling.std, s.time_sampling.data_proportion) == (-0.4, 1.0, 0.5)
        seen.add(type(s).__name__)
    assert seen == {"FlowMatchingLoss", "ImprovedMeanFlowLoss"}
    mf = create_loss_strategy(TrainFlowConfig(**_valid(loss_strategy="mean_flow", gamma=0.25, c=0.01,
                                                       noise_schedule="uniform", time_sampling="uniform")))
    assert isinstance(mf, MeanFlowLoss) and mf.gamma == 0.25 and mf.c == 0.01
    assert isinstance(mf.time_sampling, MeanFlowTimeSampling) and mf.noise_schedule.noise_max == 1.0
