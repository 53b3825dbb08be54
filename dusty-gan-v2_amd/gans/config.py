"""Minimal stand-in for the OmegaConf nodes the reference passes around (omegaconf is not in the
image): nested dict with attribute access, loadable from the same YAML files."""
import copy
import os

import yaml

DEFAULT_CONFIG = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                              "configs", "gans", "dusty_v2.yaml")


class Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return Config({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_config(obj):
    if isinstance(obj, dict):
        return Config({k: to_config(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return [to_config(v) for v in obj]
    return obj


def load_config(path=DEFAULT_CONFIG):
    with open(path) as f:
        return to_config(yaml.safe_load(f))
