"""Shared test helpers (fixtures loading, tolerances)."""
import os, json
import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    out = {}
    for k in z.files:
        v = z[k]
        out[k] = torch.from_numpy(v) if v.dtype.kind in 'fiub' else v
    return out


def golden_json(name):
    return json.load(open(os.path.join(GOLDEN, name + '.json')))


def assert_close(a, b, tol, what='', scale=None):
    """max|a-b| <= tol * max|b|  (tolerance relative to the tensor's scale)."""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, '%s: shape %s vs %s' % (what, tuple(a.shape), tuple(b.shape))
    s = scale if scale is not None else max(b.abs().max().item(), 1e-30)
    err = (a - b).abs().max().item()
    assert err <= tol * s, '%s: max err %.3e > %.1e * scale %.3e' % (what, err, tol, s)
    return err


def req(sd):
    return {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v.clone())
            for k, v in sd.items()}


def tied_grads(sdg):
    """Gradients of an oracle run keyed like the reference's named_parameters (tied q/k summed, N2)."""
    g = {}
    for k, v in sdg.items():
        if not isinstance(v, torch.Tensor) or v.grad is None or '.key.' in k:
            continue
        gg = v.grad
        if '.query.' in k:
            kk = k.replace('.query.', '.key.')
            if kk in sdg and sdg[kk].grad is not None:
                gg = gg + sdg[kk].grad
        g[k] = gg
    return g
