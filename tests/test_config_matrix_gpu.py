"""GPU: SURVEY 8(f) N1 -- the reference's experiment matrix runs through ``train_flow`` on this backend.

Every ``configs/method=*.json`` of the reference (72 files: 4 methods x 3 architectures x 2 datasets x {mdct, reshape} = the
48 tokenization entries SURVEY N1 names, plus the 24 older files without tokenization / architecture keys, which the
reference's defaults turn into MLP runs on the raw vector; the raw JSON text is in ``tests/golden/configs_reference.json``, generated from the reference's files by
``tests/golden/gen_config_golden.py``) is loaded by the config mirror and trained for two steps with the reference's
``train_flow`` flow (``trainers/train.py:156-507``): tokenise -> train_step -> log -> sample -> checkpoint, asserting
the work-directory layout each time.  What is overridden, for run time only: ``n_steps`` 2, ``batch_size`` 8,
``sample_steps`` 2, ``num_blocks`` 2, the audio ``noise_dimension`` 1024 samples (MDCT 512/256 -> 3 x 512 tokens,
reshape 128 -> 8 x 128), synthetic batches instead of ``data_dir``.  Architecture, method / loss strategy, tokenization,
condition / latent dimensions, learning rate, weight decay, seeds stay as the JSON has them."""
import json

import pytest
import torch

pytestmark = pytest.mark.gpu


def _matrix(golden_dir):
    d = json.loads((golden_dir / "configs_reference.json").read_text())
    return {k: v["raw"] for k, v in d.items() if k.startswith("method=")}


def test_matrix_is_complete(golden_dir):
    names = _matrix(golden_dir)
    assert len(names) == 72
    combos = set()
    for n, raw in names.items():
        if raw.get("tokenization_strategy") is not None:
            combos.add((raw["method"], raw["architecture"], raw["dataset"], raw["tokenization_strategy"]))
    assert len(combos) == 48                                         # the 48 method x tokenization entries of N1
    assert {c[0] for c in combos} == {"autoencoder", "flow_matching", "mean_flow", "improved_mean_flow"}
    assert {c[1] for c in combos} == {"mlp", "mlp_mixer", "convnet"}
    assert {c[2] for c in combos} == {"mnist", "audio"} and {c[3] for c in combos} == {"mdct", "reshape"}


def test_every_matrix_config_trains_two_steps(golden_dir, tmp_path):
    from meanflow_audio_codec_amd.configs import load_config_from_json
    from meanflow_audio_codec_amd.models.conv_flow import ConditionalConvFlow
    from meanflow_audio_codec_amd.models.mlp_flow import ConditionalFlow
    from meanflow_audio_codec_amd.models.mlp_mixer import ConditionalMLPMixerFlow
    from meanflow_audio_codec_amd.trainers.loss_strategies import FlowMatchingLoss, ImprovedMeanFlowLoss, MeanFlowLoss
    from meanflow_audio_codec_amd.trainers.train import create_loss_strategy, synthetic_iterator, train_flow
    arch_cls = {"mlp": ConditionalFlow, "mlp_mixer": ConditionalMLPMixerFlow, "convnet": ConditionalConvFlow}
    want_D = {("mnist", None): 784, ("mnist", "mdct"): 1024, ("mnist", "reshape"): 784,
              ("audio", None): 1024, ("audio", "mdct"): 1536, ("audio", "reshape"): 1024}
    ran = 0
    for name, raw in sorted(_matrix(golden_dir).items()):
        raw = dict(raw)
        raw.update(n_steps=2, batch_size=8, sample_steps=2, sample_every=1, num_blocks=2, checkpoint_step=None)
        if raw["noise_dimension"] > 2048:          # the audio entries (196608 samples, or 32768 in the older files)
            raw["noise_dimension"] = 1024
        wd = tmp_path / name.replace(".json", "")
        raw["workdir"] = str(wd)
        p = tmp_path / "cfg.json"
        p.write_text(json.dumps(raw))
        cfg = load_config_from_json(p)
        strategy = create_loss_strategy(cfg)
        # trainers/train.py:52-67: no loss_strategy key -> improved_mean_flow iff use_improved_mean_flow, else
        # flow_matching ("method" is metadata: autoencoder / mean_flow configs without the key train flow matching,
        # exactly as the reference's train_flow would)
        if cfg.loss_strategy is None:
            assert isinstance(strategy, ImprovedMeanFlowLoss if cfg.use_improved_mean_flow else FlowMatchingLoss), name
        else:
            assert isinstance(strategy, {"flow_matching": FlowMatchingLoss, "mean_flow": MeanFlowLoss,
                                         "improved_mean_flow": ImprovedMeanFlowLoss}[cfg.loss_strategy]), name
        state, token_shape = train_flow(cfg, synthetic_iterator(cfg))
        tk = raw.get("tokenization_strategy")
        D = want_D[(raw["dataset"], tk)] if tk is not None else raw["noise_dimension"]
        assert isinstance(state.model, arch_cls[raw.get("architecture") or "mlp"]), name
        assert state.model.noise_dimension == D, (name, state.model.noise_dimension, D)
        assert (token_shape is None) == (tk is None)
        if token_shape is not None:
            assert token_shape[0] * token_shape[1] == D
        assert state.step == 2
        for rel in ("config.json", "metadata.json", "summary.json", "logs/train_log.jsonl",
                    "checkpoints/step_00002.msgpack", "checkpoints/step_00002.json", "samples/step_0000.npy",
                    "samples/step_0001.npy", "samples/step_0002.npy"):
            assert (wd / rel).exists(), (name, rel)
        rows = [json.loads(l) for l in (wd / "logs" / "train_log.jsonl").read_text().splitlines()]
        assert [r["step"] for r in rows] == [0, 1] and all(r["loss"] == r["loss"] and abs(r["loss"]) < 1e9 for r in rows), name
        import numpy as np
        smp = np.load(wd / "samples" / "step_0002.npy")
        assert smp.shape[0] == 8 and np.isfinite(smp).all(), name
        # samples come back in the DATA domain (detokenised): noise_dimension samples / pixels per row
        per_row = int(np.prod(smp.shape[1:]))
        if raw.get("tokenization_strategy") == "mdct":
            assert per_row == (token_shape[0] - 1) * 256 + 1024, (name, smp.shape)
        else:
            assert per_row == raw["noise_dimension"], (name, smp.shape)
        for q in (wd / "checkpoints").glob("*.msgpack"):     # keep the tmp dir small
            q.unlink()
        del state
        torch.cuda.empty_cache()
        ran += 1
    assert ran == 72
