"""Seeded synthetic weights (no checkpoints are available offline).

The same routine fills a reference module (golden generation), the CPU oracle and the HIP weight packs, so
all three see bit-identical parameters.  Rule per tensor (in ``state_dict`` order, index i):
``g = manual_seed(seed*100003 + i)``; ndim>=2 -> randn * fan_in**-0.5 (* gain); norm weights -> 1 + 0.1*randn;
other 1-D (biases) -> 0.05*randn.
"""
import torch


def synth_tensor(name: str, shape, index: int, seed: int = 0, gain: float = 1.0) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(seed * 100003 + index)
    shape = tuple(shape)
    if len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return torch.randn(shape, generator=g, dtype=torch.float32) * (gain * fan_in ** -0.5)
    is_norm_w = name.endswith("weight") and ("norm" in name or ".0.weight" in name and False)
    if is_norm_w:
        return 1.0 + 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
    return 0.05 * torch.randn(shape, generator=g, dtype=torch.float32)


def synth_state_dict(names_shapes, seed: int = 0, gain: float = 1.0, norm_names=()):
    """names_shapes: iterable of (name, shape).  ``norm_names``: names of 1-D *scale* tensors (GroupNorm /
    LayerNorm weights) that must be centred on 1 rather than 0."""
    norm_names = set(norm_names)
    out = {}
    for i, (n, s) in enumerate(names_shapes):
        g = torch.Generator(device="cpu")
        g.manual_seed(seed * 100003 + i)
        s = tuple(s)
        if len(s) >= 2:
            fan_in = 1
            for d in s[1:]:
                fan_in *= d
            out[n] = torch.randn(s, generator=g, dtype=torch.float32).mul_(gain * fan_in ** -0.5)
        elif n in norm_names:
            out[n] = torch.randn(s, generator=g, dtype=torch.float32).mul_(0.1).add_(1.0)
        else:
            out[n] = torch.randn(s, generator=g, dtype=torch.float32).mul_(0.05)
    return out


def fill_module_(module: torch.nn.Module, seed: int = 0, gain: float = 1.0):
    """Fill every parameter of a torch module in state_dict order; returns the list of (name, shape)."""
    sd = module.state_dict()
    norm_names = set()
    for mname, m in module.named_modules():
        if isinstance(m, (torch.nn.GroupNorm, torch.nn.LayerNorm)):
            norm_names.add((mname + "." if mname else "") + "weight")
    names_shapes = [(k, tuple(v.shape)) for k, v in sd.items()]
    new = synth_state_dict(names_shapes, seed, gain, norm_names)
    with torch.no_grad():
        for k, v in sd.items():
            v.copy_(new[k].to(v.dtype))
    return names_shapes, sorted(norm_names)
