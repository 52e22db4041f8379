"""Shared helpers for the parity tests (synthetic data per SURVEY.md section 8d)."""
import math

import torch


def make_data(n, d, seed=0, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(n, d, generator=g, dtype=dtype)
    y = torch.sin(2 * math.pi * X[:, 0]) + torch.cos(math.pi * X.sum(-1)) + 0.1 * torch.randn(n, generator=g, dtype=dtype)
    return X, y


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))


def free_port() -> int:
    """A TCP port the OS just handed out (bind to 0) -- one per multi-process test, so tests in one pytest process
    never reuse a rendezvous port."""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]
