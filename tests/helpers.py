"""Shared test helpers: golden-fixture loading and tolerance checks."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def source_digest():
    """sha1 of the kernel sources (glow-tts-train_amd/csrc/*.hip, *.hpp) with comments and white space removed: what a
    measurement file (parity margins) was taken on.  Comment-only edits do not change it; any code edit does."""
    import hashlib
    import re

    csrc = os.path.join(ROOT, "glow-tts-train_amd", "csrc")
    h = hashlib.sha1()
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".hpp")):
            text = open(os.path.join(csrc, fn)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            text = re.sub(r"//[^\n]*", "", text)
            h.update(fn.encode())
            h.update(re.sub(r"\s+", "", text).encode())
    return h.hexdigest()[:16]


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


# ---------------------------------------------------------------------------------------------------------------------
# dropout decisions as data (ops.keep_mask_tap / keep_mask_inject  <->  oracle.KeepMasks site names)
def encoder_layer_mask_sizes(b, heads, hidden, filt, t):
    """Element counts of the four keep-masks of one transformer layer, in the order the package draws them
    (attentions.Encoder.forward): p_attn, attention branch, FFN hidden, FFN branch."""
    return [("attn", (b, heads, t, t)), ("y1", (b, hidden, t)), ("ffn", (b, filt, t)), ("y2", (b, hidden, t))]


def tapped_masks_to_oracle_sites(tapped, heads, hidden, filt):
    """[(site, uint8 CPU tensor, p)] as recorded by ops.keep_mask_tap -> {oracle site: (keep, p)} (oracle.KeepMasks)."""
    out = {}
    for site, mask, p in tapped:
        if site == "decoder.wn":                         # (blocks, layers, B, 2H, T')
            for k in range(mask.shape[0]):
                for l in range(mask.shape[1]):
                    out[f"decoder.flows.{3 * k + 2}.wn.{l}"] = (mask[k, l], p)
        elif site.startswith("decoder.block."):          # (layers, B, 2H, T') drawn by one block
            k = int(site.rsplit(".", 1)[1])
            for l in range(mask.shape[0]):
                out[f"decoder.flows.{3 * k + 2}.wn.{l}"] = (mask[l], p)
        elif site == "encoder.layers":                   # flat: layers x [attn, y1, ffn, y2] — needs (b, t)
            raise RuntimeError("encoder.layers: use split_encoder_layers()")
        else:
            out[site] = (mask, p)
    return out


def split_encoder_layers(flat, p, b, t, heads, hidden, filt, prefix="encoder.encoder"):
    sizes = encoder_layer_mask_sizes(b, heads, hidden, filt, t)
    per = sum(int(np.prod(s)) for _, s in sizes)
    assert flat.numel() % per == 0, (flat.numel(), per)
    out, pos = {}, 0
    for i in range(flat.numel() // per):
        for name, shape in sizes:
            n = int(np.prod(shape))
            out[f"{prefix}.{i}.{name}"] = (flat[pos: pos + n].view(*shape), p)
            pos += n
    return out


class MaskTap:
    """Context manager: records every keep-mask the package draws during a step (copied to the host)."""

    def __init__(self, ops):
        self.ops, self.rec = ops, []

    def __enter__(self):
        assert self.ops.keep_mask_tap is None
        self.ops.keep_mask_tap = lambda site, mask, p: self.rec.append((site, mask.detach().cpu(), p))
        return self

    def __exit__(self, *exc):
        self.ops.keep_mask_tap = None

    def oracle_sites(self, b, t_text, heads, hidden, filt):
        out = {}
        for site, mask, p in self.rec:
            if site == "encoder.layers":
                out.update(split_encoder_layers(mask, p, b, t_text, heads, hidden, filt))
            else:
                out.update(tapped_masks_to_oracle_sites([(site, mask, p)], heads, hidden, filt))
        return out


class MaskInject:
    """Context manager: makes the package use recorded keep decisions ({oracle site: (keep uint8 array, p)})."""

    def __init__(self, ops, sites, n_blocks, n_block_layers, n_enc, device="cuda"):
        self.ops, self.sites, self.dev = ops, sites, device
        self.n_blocks, self.n_block_layers, self.n_enc = n_blocks, n_block_layers, n_enc
        self.served = set()

    def _get(self, name, p):
        keep, p_rec = self.sites[name]
        assert abs(p_rec - p) < 1e-9, (name, p_rec, p)
        self.served.add(name)
        return torch.as_tensor(np.asarray(keep), dtype=torch.uint8)

    def _inject(self, site, shape, p):
        if site == "decoder.wn":
            m = torch.stack([torch.stack([self._get(f"decoder.flows.{3 * k + 2}.wn.{l}", p) for l in range(self.n_block_layers)])
                             for k in range(self.n_blocks)])
        elif site == "encoder.layers":
            m = torch.cat([self._get(f"encoder.encoder.{i}.{n}", p).reshape(-1) for i in range(self.n_enc)
                           for n in ("attn", "y1", "ffn", "y2")])
        elif site.startswith("decoder.block."):          # one block's (layers, B, 2H, T') drawn by the block itself
            k = int(site.rsplit(".", 1)[1])
            m = torch.stack([self._get(f"decoder.flows.{3 * k + 2}.wn.{l}", p) for l in range(self.n_block_layers)])
        elif site in self.sites:
            m = self._get(site, p)
        else:
            raise AssertionError(f"keep-mask drawn for an unknown site {site!r} {shape}")
        assert tuple(m.shape) == tuple(shape), (site, tuple(m.shape), shape)
        return m.contiguous().to(self.dev)

    def __enter__(self):
        assert self.ops.keep_mask_inject is None
        self.ops.keep_mask_inject = self._inject
        return self

    def __exit__(self, *exc):
        self.ops.keep_mask_inject = None
