"""Classification heads on the HIP path (reference: headers/margin_losses.py).

Softmax (:18-68), AMCosFace (:203-315), AMArcFace (:318-428): same constructor signatures,
same `weight` / `bias` parameters (xavier-uniform / zeros), forward(embedding, label) returns
the (B, C) logits; normalisation, GEMM and margin run in libmsml_hip.so."""
import torch
from torch import nn
from torch.nn import Parameter

from .. import functional as Fh
from .._lib import BF16, F32

__all__ = ["Softmax", "AMCosFace", "AMArcFace"]


def _mode(m):
    return BF16 if getattr(m, "fp16", False) else F32


class Softmax(nn.Module):
    def __init__(self, in_features, out_features, device_id):
        super().__init__()
        self.in_features, self.out_features, self.device_id = in_features, out_features, device_id
        self.weight = Parameter(torch.FloatTensor(out_features, in_features))
        self.bias = Parameter(torch.FloatTensor(out_features))
        nn.init.xavier_uniform_(self.weight)
        nn.init.zeros_(self.bias)

    def forward(self, embedding, label):
        if self.device_id is not None:
            raise ValueError("DataParallel is not implemented yet.")
        return Fh.linear(embedding, self.weight, self.bias, _mode(self))


class _AMHead(nn.Module):
    kind = "arc"

    def __init__(self, in_features, out_features, device_id, s, m, a, k):
        super().__init__()
        print("%s, s=%.1f, m=%.2f, a=%.2f, k=%.2f" % (type(self).__name__, s, m, a, k))
        self.in_features, self.out_features, self.device_id = in_features, out_features, device_id
        self.s, self.m, self.a, self.k = s, m, a, k
        self.weight = Parameter(torch.FloatTensor(out_features, in_features))
        nn.init.xavier_uniform_(self.weight)

    def forward(self, embedding, label):
        if self.device_id is not None:
            raise ValueError("DataParallel is not implemented yet.")
        return Fh.cos_margin_head(embedding, self.weight, label, self.kind, float(self.s),
                                  float(self.m), float(self.a), float(self.k), _mode(self))

    def __repr__(self):
        return "%s(in_features = %s, out_features = %s, s = %s, m = %s, a = %s, k = %s)" % (
            type(self).__name__, self.in_features, self.out_features, self.s, self.m, self.a, self.k)


class AMCosFace(_AMHead):
    kind = "cos"

    def __init__(self, in_features, out_features, device_id, s=64.0, m=0.4, a=1.2, k=0.1):
        super().__init__(in_features, out_features, device_id, s, m, a, k)


class AMArcFace(_AMHead):
    kind = "arc"

    def __init__(self, in_features, out_features, device_id, s=64.0, m=0.5, a=1.2, k=0.1):
        super().__init__(in_features, out_features, device_id, s, m, a, k)
