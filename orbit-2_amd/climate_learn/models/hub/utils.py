"""Model registry (same contract as the reference's models/hub/utils.py:1-9)."""
MODEL_REGISTRY = {}


def register(name):
    def deco(cls):
        MODEL_REGISTRY[name] = cls
        return cls
    return deco
