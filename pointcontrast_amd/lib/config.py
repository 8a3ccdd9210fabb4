"""Configuration tree of the pre-training run.

The reference drives pc/ddp_train.py with Hydra/OmegaConf (pc/config/defaults.yaml); neither
is installed here, so this is a small attribute-access tree with the same groups and keys
(trainer / net / opt / misc / data), the same defaults, YAML loading and Hydra-style
``group.key=value`` overrides."""
import copy

import yaml

# values restated from pc/config/defaults.yaml:4-86 (hot-path keys only)
DEFAULTS = {
    "trainer": {
        "trainer": "HardestContrastiveLossTrainer", "batch_size": 4,
        "num_pos_per_batch": 1024, "num_hn_samples_per_batch": 256,
        "neg_thresh": 1.4, "pos_thresh": 0.1,
        "use_random_scale": False, "min_scale": 0.8, "max_scale": 1.2,
        "use_random_rotation": True, "rotation_range": 360,
        "stat_freq": 40, "lr_update_freq": 1000, "positive_pair_search_voxel_size_multiplier": 1.5,
    },
    "net": {"model": "Res16UNet34C", "model_n_out": 32, "conv1_kernel_size": 3, "normalize_feature": True},
    "opt": {"optimizer": "SGD", "max_iter": 300000, "lr": 0.1, "momentum": 0.8, "weight_decay": 1e-4,
            "bn_momentum": 0.05, "exp_gamma": 0.99, "scheduler": "ExpLR"},
    "misc": {"out_dir": "./outputs", "use_gpu": True, "num_gpus": 1, "weight": None, "lenient_weight_loading": False,
             "weight_kernel_order": "hybrid",  # "hypercube": see lib/checkpoint.py
             "train_num_thread": 2, "nceT": 0.07, "npos": 4096, "seed": 0},
    "data": {"dataset": "SyntheticScanNetPairDataset", "voxel_size": 0.025, "dataset_root_dir": None,
             "scannet_match_dir": None, "num_pairs": 64},
}


class Config(dict):
  """dict with attribute access, nested."""

  def __getattr__(self, k):
    try:
      return self[k]
    except KeyError:
      raise AttributeError(k)

  def __setattr__(self, k, v):
    self[k] = v

  @staticmethod
  def wrap(d):
    return Config({k: Config.wrap(v) if isinstance(v, dict) else v for k, v in d.items()})

  def to_dict(self):
    return {k: v.to_dict() if isinstance(v, Config) else v for k, v in self.items()}


def _parse(v):
  try:
    return yaml.safe_load(v)
  except Exception:
    return v


def get_config(overrides=(), yaml_path=None):
  """defaults <- optional YAML file <- ``a.b=c`` overrides (the CLI form of scripts/ddp_local.sh)."""
  tree = copy.deepcopy(DEFAULTS)
  if yaml_path:
    with open(yaml_path) as f:
      loaded = yaml.safe_load(f) or {}
    for g, kv in loaded.items():
      if isinstance(kv, dict):
        tree.setdefault(g, {}).update(kv)
  for ov in overrides:
    key, _, val = ov.partition("=")
    parts = key.split(".")
    node = tree
    for p in parts[:-1]:
      node = node.setdefault(p, {})
    node[parts[-1]] = _parse(val)
  return Config.wrap(tree)
