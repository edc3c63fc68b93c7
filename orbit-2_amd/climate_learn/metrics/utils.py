"""Metric registry + meta-info record (same contract as the reference's metrics/utils.py)."""
from dataclasses import dataclass
from typing import Any, List

METRICS_REGISTRY = {}


@dataclass
class MetricsMetaInfo:
    in_vars: List[str]
    out_vars: List[str]
    lat: Any
    lon: Any
    climatology: Any


def register(name):
    def deco(cls):
        METRICS_REGISTRY[name] = cls
        cls.name = name
        return cls
    return deco
