"""Shared test helpers: golden-fixture loading and tolerance checks."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: d[k] for k in d.files}


def split_prefix(d, prefix, as_torch=True, requires_grad=False):
    out = {}
    for k, v in d.items():
        if k.startswith(prefix):
            t = torch.from_numpy(np.array(v)) if as_torch else v
            if as_torch and requires_grad and t.is_floating_point():
                t.requires_grad_(True)
            out[k[len(prefix):]] = t
    return out


def T(a, **kw):
    return torch.from_numpy(np.array(a)).to(**kw)


def assert_close(actual, expected, rtol=1e-5, atol=1e-5, what=""):
    a = actual.detach().cpu().double().numpy() if isinstance(actual, torch.Tensor) else np.asarray(actual, np.float64)
    e = expected.detach().cpu().double().numpy() if isinstance(expected, torch.Tensor) else np.asarray(expected, np.float64)
    assert a.shape == e.shape, f"{what}: shape {a.shape} vs {e.shape}"
    err = np.abs(a - e)
    tol = atol + rtol * np.abs(e)
    if not (err <= tol).all():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{what}: max abs err {err.max():.3e} at {i}: got {a[i]!r} want {e[i]!r} "
                             f"(rtol={rtol}, atol={atol})")


def rel_err(actual, expected):
    """||a-e||_inf / max(||e||_inf, tiny): the north-star's '1e-3 rel' metric for a whole tensor."""
    a = actual.detach().cpu().double() if isinstance(actual, torch.Tensor) else torch.as_tensor(actual).double()
    e = expected.detach().cpu().double() if isinstance(expected, torch.Tensor) else torch.as_tensor(expected).double()
    return float((a - e).abs().max() / e.abs().max().clamp_min(1e-12))
