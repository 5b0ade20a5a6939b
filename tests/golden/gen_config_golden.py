"""Generate config golden data from the reference's own (stdlib-only) config module.

    python tests/golden/gen_config_golden.py

Loads ``/root/reference/meanflow_audio_codec/configs/config.py`` by file path and records, for every
``/root/reference/configs/*.json``, the raw JSON and the reference's ``TrainFlowConfig.to_dict()``.
Only this data file is committed (the JSONs are the reference's experiment matrix = input data).
"""
import importlib.util
import json
import pathlib
import sys

REF = pathlib.Path("/root/reference")
spec = importlib.util.spec_from_file_location("ref_config", REF / "meanflow_audio_codec/configs/config.py")
mod = importlib.util.module_from_spec(spec)
sys.modules["ref_config"] = mod
spec.loader.exec_module(mod)

out = {}
for p in sorted((REF / "configs").glob("*.json")):
    raw = json.loads(p.read_text())
    try:
        cfg = mod.load_config_from_json(p)
        d = cfg.to_dict()
        flat = {k: getattr(cfg, k) for k in ("batch_size", "n_steps", "base_lr", "weight_decay", "seed",
                                             "noise_dimension", "condition_dimension", "latent_dimension",
                                             "num_blocks", "architecture", "dataset", "tokenization_strategy",
                                             "method", "use_improved_mean_flow", "loss_strategy", "sample_steps")}
        out[p.name] = {"raw": raw, "to_dict": json.loads(json.dumps(d, default=str)), "flat": flat}
    except Exception as e:  # record what the reference rejects, too
        out[p.name] = {"raw": raw, "error": f"{type(e).__name__}: {e}"}
dst = pathlib.Path(__file__).parent / "configs_reference.json"
dst.write_text(json.dumps(out, indent=0, sort_keys=True))
print(len(out), "configs ->", dst, sum("error" in v for v in out.values()), "rejected")
