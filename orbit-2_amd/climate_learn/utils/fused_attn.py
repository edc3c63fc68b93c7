"""Attention-backend switch -- the reference's own operator plug point (utils/fused_attn.py:13-16 there).

CK / DEFAULT / NONE keep the reference's *semantics* (which of them apply P-dropout in eval mode, see
components/attention.py) but all of them execute the hand-written HIP flash-attention kernels; HIP is the
native name and the default on gfx950."""
import enum


class FusedAttn(enum.Enum):
    HIP = "HIP"          # MI355X-native kernels, dropout in training mode only
    CK = "CK"            # reference semantics: P-dropout is applied even in eval() (attention.py:57 there)
    DEFAULT = "DEFAULT"  # dropout in training mode only (attention.py:69 there)
    NONE = "NONE"        # dropout in training mode only (attention.py:76 there)

    @property
    def dropout_in_eval(self) -> bool:
        return self is FusedAttn.CK
