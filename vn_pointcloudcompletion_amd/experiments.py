"""Config / experiment-directory helpers with the reference's behaviour (/root/reference/utils/experiments.py:17-171,
utils/logging_config.py): ``config.json`` -> nested SimpleNamespace, ``OUTPUT_DIR/<MM-DD>_<name>_<NNN>/{config.json,
models/, optimizer/, visualizations/, train/, val/, train.log}``.  ``OUTPUT_DIR`` / ``DATA_PATH`` come from the
environment (the reference reads them from ``.env`` via python-dotenv, main.py:10; ``load_env`` parses that file)."""
from __future__ import annotations

import json
import logging
import os
from datetime import datetime
from shutil import copy
from types import SimpleNamespace

LOGGING_CONFIG = {
    "version": 1,
    "disable_existing_loggers": False,
    "formatters": {"standard": {"format": "%(asctime)s - %(name)s - %(levelname)s - %(message)s"}},
    "handlers": {"console": {"class": "logging.StreamHandler", "level": "DEBUG", "formatter": "standard",
                             "stream": "ext://sys.stdout"}},
    "loggers": {"": {"handlers": ["console"], "level": "INFO", "propagate": False}},
}


def load_env(path=".env"):
    """KEY="value" lines -> os.environ (only keys not already set), like dotenv.load_dotenv."""
    if os.path.exists(path):
        with open(path) as f:
            for line in f:
                line = line.strip()
                if not line or line.startswith("#") or "=" not in line:
                    continue
                k, v = line.split("=", 1)
                os.environ.setdefault(k.strip(), v.strip().strip('"').strip("'"))
    os.environ.setdefault("OUTPUT_DIR", "./experiments/")
    os.environ.setdefault("DATA_PATH", "./data/")


def output_dir():
    return os.environ.get("OUTPUT_DIR", "./experiments/")


def load_config(experiment_name=None, config_path=None):
    """utils/experiments.py:17-49."""
    if config_path is None:
        if experiment_name is None:
            config_path = "./config.json"
        else:
            config_path = os.path.join(os.getcwd(), output_dir(), experiment_name, "config.json")
    with open(config_path, "r") as f:
        config = json.load(f, object_hook=lambda d: SimpleNamespace(**d))
    config.data_path = os.environ.get("DATA_PATH", "./data/")
    return config


def config_to_dict(config):
    out = {}
    for k, v in vars(config).items():
        out[k] = config_to_dict(v) if isinstance(v, SimpleNamespace) else v
    return out


def store_config(config):
    """utils/experiments.py:90-100."""
    with open(os.path.join(config.exp_dir, "config.json"), "w") as f:
        json.dump(config_to_dict(config), fp=f, indent=4, sort_keys=True)


def setup_loggers(command, log_path=None, logger_names=None):
    """utils/experiments.py:52-87."""
    if log_path is None:
        log_path = f"./logs/{command}/{datetime.now().strftime('%Y-%m-%d_%H-%M-%S')}.log"
        os.makedirs(os.path.dirname(log_path), exist_ok=True)
    handler = logging.FileHandler(filename=log_path)
    handler.setLevel("DEBUG")
    handler.setFormatter(logging.Formatter(fmt="%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
    for name in (logger_names or [command]):
        logging.getLogger(name).addHandler(handler)


def create_experiment(config):
    """utils/experiments.py:103-171: next free ``<MM-DD>_<name>_<NNN>`` directory with its sub-directories."""
    out = output_dir()
    experiment_name = f"{datetime.now().strftime('%m-%d')}_{config.name}"
    os.makedirs(out, exist_ok=True)
    prev = [int(x[len(experiment_name) + 1:]) for x in os.listdir(out)
            if x[:-4] == experiment_name and x[len(experiment_name) + 1:].isdigit()]
    num = max(prev) + 1 if prev else 0
    run_name = f"{experiment_name}_{num:03d}"
    config.run_name = run_name
    exp_dir = os.path.join(out, run_name)
    for sub in ("", "models", "optimizer", "visualizations", "train", "val"):
        os.makedirs(os.path.join(exp_dir, sub), exist_ok=True)
    if getattr(config, "checkpoint", 0) != 0:
        old = config.exp_dir
        copy(os.path.join(old, "models", f"model_{config.checkpoint}.pth"), os.path.join(exp_dir, "models"))
        copy(os.path.join(old, "optimizer", f"optim_{config.checkpoint}.pth"), os.path.join(exp_dir, "optimizer"))
    config.exp_dir = exp_dir
    return config


def get_num_params_total(model):
    return sum(p.numel() for p in model.parameters())


class ScalarWriter:
    """Stand-in for tensorboardX.SummaryWriter.add_scalar (train.py:39-40,184-186): one JSON line per scalar."""

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self._f = open(os.path.join(logdir, "scalars.jsonl"), "a")

    def add_scalar(self, tag, value, step):
        self._f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")
        self._f.flush()

    def close(self):
        self._f.close()
