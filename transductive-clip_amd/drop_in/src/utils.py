"""Helpers the method classes and the task-batch loops need, under the reference's names
(reference: src/utils.py).  Only what the EM-Dirichlet path touches is provided; CLIP feature
extraction and datasets are out of scope."""
import copy
import logging
import os
from ast import literal_eval

import numpy as np


def compute_confidence_interval(data, axis=0):
    """Mean and 95 % confidence half-width (reference: src/utils.py:27-37)."""
    a = 1.0 * np.array(data)
    m = np.mean(a, axis=axis)
    std = np.std(a, axis=axis)
    pm = 1.96 * (std / np.sqrt(a.shape[axis]))
    return m, pm


class Logger:
    """File + stderr logger with per-instance handlers, removed by del_logger()
    (reference: src/utils.py:171-221).  log_file=None logs to stderr only."""

    def __init__(self, name, log_file=None, level=logging.INFO):
        self.logger = logging.getLogger(f"{name}.{id(self)}")
        self.logger.setLevel(level)
        self.logger.propagate = False
        self._handlers = []
        fmt = logging.Formatter("[%(name)s]: [%(levelname)s]: %(message)s")
        targets = [logging.StreamHandler()]
        if log_file:
            try:
                targets.append(logging.FileHandler(log_file))
            except OSError:
                pass
        for h in targets:
            h.setFormatter(fmt)
            self.logger.addHandler(h)
            self._handlers.append(h)

    def info(self, msg):
        self.logger.info(msg)

    def warning(self, msg):
        self.logger.warning(msg)

    def del_logger(self):
        for h in self._handlers:
            self.logger.removeHandler(h)
            try:
                h.close()
            except Exception:
                pass
        self._handlers = []


class CfgNode(dict):
    """Flat attribute dict, the shape of the reference's merged config (src/utils.py:40-88)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = value


# ---- configuration files (reference: src/utils.py:90-168, main.py:19-35) -------------------------------
def _decode_cfg_value(v):
    """Command-line values arrive as strings: anything Python can read as a literal becomes that literal
    ('5' -> 5, 'True' -> True, '[1, 2]' -> [1, 2]), everything else stays a string."""
    if not isinstance(v, str):
        return v
    try:
        return literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def _check_and_coerce_cfg_value_type(replacement, original, key, full_key):
    """An override must have the type of the value it replaces; tuples and lists convert into each other."""
    if type(replacement) is type(original):
        return replacement
    for src, dst in ((tuple, list), (list, tuple)):
        if type(replacement) is src and type(original) is dst:
            return dst(replacement)
    raise ValueError("Type mismatch ({} vs. {}) with values ({} vs. {}) for config key: {}".format(
        type(original), type(replacement), original, replacement, full_key))


def load_cfg_from_cfg_file(file):
    """The reference's YAML files hold one level of sections (EVAL:, DATA:, METHOD:, ...) whose keys are
    flattened into one CfgNode; a later section overwrites an earlier one."""
    import yaml
    if not (os.path.isfile(file) and file.endswith('.yaml')):
        raise AssertionError('{} is not a yaml file'.format(file))
    with open(file, 'r') as f:
        sections = yaml.safe_load(f)
    flat = {}
    for section in sections:
        flat.update(sections[section])
    return CfgNode(flat)


def merge_cfg_from_list(cfg, cfg_list):
    """`--opts key value key value ...`: only the last dotted component of a key counts; a key the config
    already has must keep its type, an unknown key is simply added."""
    if len(cfg_list) % 2:
        raise AssertionError(cfg_list)
    merged = copy.deepcopy(cfg)
    for full_key, raw in zip(cfg_list[0::2], cfg_list[1::2]):
        key = full_key.split('.')[-1]
        value = _decode_cfg_value(raw)
        if key in cfg:
            value = _check_and_coerce_cfg_value_type(value, cfg[key], key, full_key)
        merged[key] = value
    return merged


def load_merged_config(config_root, opts=None):
    """main.py:19-35: main_config.yaml, then --opts (they choose the dataset and the method), then
    datasets_config/config_<dataset>.yaml and methods_config/<method>.yaml on top, then --opts once more so
    that the command line wins over all three files; n_class = num_classes_test."""
    cfg = load_cfg_from_cfg_file(os.path.join(config_root, 'main_config.yaml'))
    if opts:
        cfg = merge_cfg_from_list(cfg, opts)
    cfg.update(load_cfg_from_cfg_file(os.path.join(config_root, 'datasets_config', 'config_{}.yaml'.format(cfg.dataset))))
    cfg.update(load_cfg_from_cfg_file(os.path.join(config_root, 'methods_config', '{}.yaml'.format(cfg.method))))
    if opts:
        cfg = merge_cfg_from_list(cfg, opts)
    cfg.n_class = cfg.num_classes_test
    return cfg

