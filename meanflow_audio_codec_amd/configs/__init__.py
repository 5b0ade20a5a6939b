from .config import (TrainFlowConfig, create_audio_config, create_mnist_config, diff_configs,  # noqa: F401
                     load_config_from_json, merge_configs, migrate_config_v1_to_v2)
