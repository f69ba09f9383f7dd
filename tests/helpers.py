"""Shared helpers for the parity tests (same checksum/pick as oracle/make_golden.py)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def checksum(t):
    t = t.detach().double().cpu()
    return np.array([t.sum().item(), t.abs().sum().item(), t.abs().max().item()], np.float64)


def pick(t, n=64):
    f = t.detach().reshape(-1).cpu()
    # (f32 linspace is exact below 2**24 elements -- every golden; f64 beyond, where f32 would round
    # the last index past the end)
    idx = torch.linspace(0, f.numel() - 1, n, dtype=torch.float32 if f.numel() < 2 ** 24 else torch.float64).long()
    return f[idx].float().numpy()


def rel_err(a, b):
    """Norm-wise relative error ||a-b|| / ||b||."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def assert_cs(t, cs, tol, what=""):
    """Checksum comparison: abs-sum and max-abs within tol (relative); plain sum within
    tol * abs-sum (it cancels)."""
    got = checksum(t)
    assert abs(got[1] - cs[1]) <= tol * abs(cs[1]) + 1e-12, (what, got, cs)
    assert abs(got[2] - cs[2]) <= tol * abs(cs[2]) + 1e-12, (what, got, cs)
    assert abs(got[0] - cs[0]) <= tol * abs(cs[1]) + 1e-12, (what, got, cs)
