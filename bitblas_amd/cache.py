"""Operator cache with the reference's interface (bitblas/cache/operator.py:24-203).

Upstream the cache exists to avoid minutes of JIT + tuning: it maps a frozen config to a built
operator and persists `{config}.json`, `mapping.json`, the TVM module tarball, the generated
wrapper source and the compiled wrapper under `~/.cache/bitblas/<arch>/<sha256(repr(config))>/`.
With a static kernel library there is nothing to compile, so the database keeps only the two JSON
files (same directory scheme, same hash) - enough for `load_from_database` to re-create the
operators and for tools that enumerate the database.
"""
from __future__ import annotations

import json
import logging
import os
import tempfile
import threading
from dataclasses import asdict
from hashlib import sha256

logger = logging.getLogger(__name__)

BITBLAS_DEFAULT_CACHE_PATH = os.path.expanduser("~/.cache/bitblas")       # bitblas/common.py:6
BITBLAS_DATABASE_PATH = os.path.expanduser(os.environ.get("BITBLAS_DATABASE_PATH", "~/.cache/bitblas"))


def _arch_dir(target) -> str:
    return str(target).replace(" ", "_").replace("=", "-").replace("/", "_")


class OperatorCache:
    # re-entrant: load_from_database -> _load_operator -> add (upstream uses an RLock too)
    cache_locker = threading.RLock()

    def __init__(self):
        self.cache = {}

    def add(self, config, op_inst):
        with self.cache_locker:
            self.cache[config] = op_inst

    def get(self, config):
        with self.cache_locker:
            return self.cache.get(config)

    def exists(self, config):
        return config in self.cache

    def clear(self):
        with self.cache_locker:
            self.cache.clear()

    def size(self):
        return len(self.cache)

    @staticmethod
    def config_hash(config) -> str:
        return sha256(repr(config).encode()).hexdigest()

    def save_into_database(self, database_path=None, target=None):
        with self.cache_locker:
            if database_path is None:
                database_path = tempfile.mkdtemp()
            os.makedirs(database_path, exist_ok=True)
            for config, op_inst in self.cache.items():
                arch_path = os.path.join(database_path, _arch_dir(target or op_inst.target))
                config_path = os.path.join(arch_path, self.config_hash(config))
                if os.path.exists(config_path):
                    continue
                os.makedirs(config_path, exist_ok=True)
                config_type, operator_type = type(config).__name__, type(op_inst).__name__
                with open(os.path.join(config_path, f"{config_type}.json"), "w") as f:
                    json.dump(asdict(config), f)
                with open(os.path.join(config_path, "mapping.json"), "w") as f:
                    # "tuned": what hardware_aware_finetune measured on the device (the reference stores its tuned hint as the
                    # generated source; here it is one integer of the descriptor)
                    tuned = {"two_pass_min_m": int(getattr(getattr(op_inst, "_desc", None), "two_pass_min_m", 0) or 0)}
                    json.dump({"config_type": config_type, "operator_type": operator_type, "tuned": tuned}, f)
                with open(os.path.join(config_path, "source.txt"), "w") as f:
                    f.write(op_inst.get_source())
            return database_path

    def load_from_database(self, database_path, target=None):
        with self.cache_locker:
            if not os.path.exists(database_path):
                logger.info("Database path %s does not exist, skipping", database_path)
                return
            arch_path = os.path.join(database_path, _arch_dir(target))
            if not os.path.exists(arch_path):
                logger.info("Target %s does not exist in the database, skipping", target)
                return
            for entry in sorted(os.listdir(arch_path)):
                config_path = os.path.join(arch_path, entry)
                if os.path.isdir(config_path):
                    self._load_operator(config_path, target)

    def _load_operator(self, config_path, target):
        import bitblas_amd
        mapping = config = None
        for name in os.listdir(config_path):
            full = os.path.join(config_path, name)
            if name == "mapping.json":
                with open(full) as f:
                    mapping = json.load(f)
            elif name.endswith(".json"):
                with open(full) as f:
                    config = json.load(f)
        if not (mapping and config):
            return
        # the default database path (~/.cache/bitblas) is shared with upstream BitBLAS: an entry of an operator
        # or config type this build does not have, or with fields it does not know, is skipped, never fatal
        try:
            config_cls = getattr(bitblas_amd, mapping["config_type"])
            operator_cls = getattr(bitblas_amd, mapping["operator_type"])
            if isinstance(config.get("M"), list):
                config["M"] = tuple(config["M"])
            cfg = config_cls(**config)
            op = operator_cls(config=cfg, target=target, enable_tuning=False, from_database=True)
            tuned = mapping.get("tuned") or {}
            if tuned.get("two_pass_min_m") and hasattr(op, "_desc"):
                op._desc.two_pass_min_m = int(tuned["two_pass_min_m"])
                op.plans = {m: op.lib.plan(m) for m in op.plans}
        except Exception as exc:  # an entry written by another build that we cannot serve
            logger.warning("skipping database entry %s: %s", config_path, exc)
            return
        self.add(cfg, op)


global_operator_cache = OperatorCache()


def load_global_ops_cache(database_path=None, target=None):
    from .target import auto_detect_nvidia_target
    database_path = database_path or get_database_path()
    target = target or auto_detect_nvidia_target()
    global_operator_cache.load_from_database(database_path, target)
    return global_operator_cache


def get_database_path():
    return BITBLAS_DATABASE_PATH


def set_database_path(path):
    global BITBLAS_DATABASE_PATH
    BITBLAS_DATABASE_PATH = path
    return BITBLAS_DATABASE_PATH
