"""Deterministic, name-hashed synthetic weights.

There is no network: neither the EfficientNet-advprop / I3D-ImageNet checkpoints nor any
dataset can be fetched, so benchmarks, smoke tests and golden fixtures all run from weights
that any machine can regenerate from the parameter *names and shapes* alone.  The statistics
follow the reference initialisation (networks/segtran_shared.py:1246-1256 normal(0, 0.02) for
Linear, :538-546 / :392-402 identity bias) closely enough that activations stay O(1) through
32 MBConv blocks / the Inception stack and attention scores stay in a realistic range.
"""
import math
import zlib
import torch


def _seed(name):
    return zlib.crc32(name.encode('utf-8')) & 0x7FFFFFFF


def _randn(name, shape):
    g = torch.Generator(device='cpu')
    g.manual_seed(_seed(name))
    return torch.randn(tuple(shape), generator=g, dtype=torch.float32, device='cpu')


def _canonical(name):
    # N2: key and query are one tied Parameter in the reference; state_dict stores both names.
    return name.replace('.key.weight', '.query.weight').replace('.key.bias', '.query.bias')


def synth_tensor(name, shape):
    """Value for state_dict entry `name` of `shape` (fp32, CPU)."""
    shape = tuple(shape)
    leaf = name.rsplit('.', 1)[-1]
    cname = _canonical(name)
    if leaf == 'num_batches_tracked':
        return torch.zeros(shape, dtype=torch.long, device='cpu')
    if leaf == 'running_mean':
        return 0.1 * _randn(cname, shape)
    if leaf == 'running_var':
        return 1.0 + 0.1 * _randn(cname, shape).abs()
    if leaf == 'attractors':
        return _randn(cname, shape)
    is_norm = any(t in name for t in ('norm_layer', '_bn', '.bn.', '_gn', 'norm_layers'))
    if is_norm:
        return (1.0 + 0.1 * _randn(cname, shape)) if leaf == 'weight' else 0.1 * _randn(cname, shape)
    if leaf == 'bias':
        return 0.02 * _randn(cname, shape)
    if leaf == 'weight' and len(shape) >= 2:
        in_transformer = 'voxel_fusion' in name
        if in_transformer and 'pos_fc' not in name:
            w = 0.02 * _randn(cname, shape)
            if '.query.' in cname or '.first_linear.' in cname:
                # identity bias on the first mode (segtran_shared.py:538-546, 392-402)
                if '.query.' in cname:
                    modes = 1 if '.in_ator_trans.' in cname else 4
                    d = shape[0] // modes
                    eye = torch.eye(d, device='cpu').repeat(1, shape[1] // d) * 0.2
                    w[:d] = w[:d] * 0.5 + eye
                else:
                    modes = 1 if '.in_ator_trans.' in cname else 4
                    f = shape[0] // modes
                    if shape[1] >= f:
                        w[:f, :f] = w[:f, :f] * 0.5 + torch.eye(f, device='cpu') * 0.2
            return w
        if 'pos_fc' in name:
            return _randn(cname, shape)                         # phases spread over several periods
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        gain = math.sqrt(2.0) if ('backbone' in name and 'se_' not in name) else 1.0
        return gain / math.sqrt(fan_in) * _randn(cname, shape)
    return 0.02 * _randn(cname, shape)


def synth_state_dict(shapes):
    """shapes: mapping name -> shape (e.g. {k: v.shape for k, v in model.state_dict().items()})."""
    return {k: synth_tensor(k, s) for k, s in shapes.items()}


def load_synth(model):
    """Fill `model` (any nn.Module) in place with synthetic weights; returns the state dict used."""
    cur = model.state_dict()
    derived = [k for k in cur if '.pos_coder.all_' in k]      # the reference's SlidingPosBiases index buffers: keep as built
    sd = synth_state_dict({k: tuple(v.shape) for k, v in cur.items() if k not in derived})
    model.load_state_dict(sd, strict=not derived)
    return sd


# ---------------------------------------------------------------------------------------
# Synthetic inputs; used by bench.py, smoke(), tests and tests/golden/make_golden.py
# ---------------------------------------------------------------------------------------
def sample(t, n=4096):
    """Deterministic strided sample of a (big) tensor: flatten()[::stride][:n]."""
    f = t.detach().reshape(-1)
    stride = max(1, f.numel() // n)
    return f[::stride][:n].clone()


def synth_image2d(B, S, seed=1337, S2=None):
    g = torch.Generator(device='cpu'); g.manual_seed(seed)
    return torch.randn(B, 3, S, S2 or S, generator=g, device='cpu')


def synth_fundus_mask(B, S, seed=1338):
    """uint8 {0,255} [B,3,S,S] in the loaders' on-disk encoding: ch0 = optic-disc region (incl. cup),
    ch1 = cup (nested disc), ch2 = 0."""
    g = torch.Generator(device='cpu'); g.manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(S, device='cpu'), torch.arange(S, device='cpu'), indexing='ij')
    m = torch.zeros(B, 3, S, S, dtype=torch.uint8, device='cpu')
    for b in range(B):
        c = (torch.rand(2, generator=g, device='cpu') * 0.3 + 0.35) * S
        r = (torch.rand(1, generator=g, device='cpu') * 0.1 + 0.25) * S
        d2 = (yy - c[0]) ** 2 + (xx - c[1]) ** 2
        m[b, 0][d2 <= r * r] = 255
        m[b, 1][d2 <= (0.5 * r) ** 2] = 255
    return m


def synth_brats(B, H, W, D, seed=1337, margin=8):
    """BraTS-like volume [B,4,H,W,D] with an exact-zero margin on every face (z-scored background
    is exactly 0, so get_mask is non-trivial) + integer labels [B,H,W,D] in {0..3}."""
    g = torch.Generator(device='cpu'); g.manual_seed(seed)
    x = torch.randn(B, 4, H, W, D, generator=g, device='cpu')
    md = margin if D > 2 * margin else D // 4
    x[:, :, :margin] = 0; x[:, :, -margin:] = 0
    x[:, :, :, :margin] = 0; x[:, :, :, -margin:] = 0
    x[..., :md] = 0; x[..., -md:] = 0
    lab = torch.randint(0, 4, (B, H, W, D), generator=g, device='cpu')
    return x, lab
