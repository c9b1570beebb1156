"""Packaged default configuration files and loaders."""
import configparser
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def load(name, **overrides):
    """Read configs/<name>.config into a RawConfigParser; overrides are {'section.key': value}."""
    cfg = configparser.RawConfigParser()
    path = os.path.join(_HERE, name + ".config")
    if not cfg.read(path):
        raise FileNotFoundError(path)
    for k, v in overrides.items():
        sec, key = k.split(".", 1)
        cfg.set(sec, key, str(v))
    return cfg


def env_config(**overrides):
    return load("env", **overrides)


def policy_config(**overrides):
    return load("policy", **overrides)
