"""Helpers to read the committed golden fixtures (tests/golden/*.npz, made by tests/golden/gen_golden.py)."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def rel_l2(a, b):
    a = torch.as_tensor(a).detach().double().cpu().reshape(-1)
    b = torch.as_tensor(b).detach().double().cpu().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def packed_error(data, prefix, t):
    """Relative L2 error of tensor ``t`` against a fixture entry written by gen_golden.pack()."""
    t = torch.as_tensor(t).detach().to(torch.float32).cpu().contiguous()
    if prefix in data:
        ref = torch.from_numpy(data[prefix])
        assert tuple(ref.shape) == tuple(t.shape), (prefix, ref.shape, t.shape)
        return rel_l2(t, ref)
    stride = int(data[prefix + "::stride"])
    assert tuple(int(s) for s in data[prefix + "::shape"]) == tuple(t.shape), (prefix, data[prefix + "::shape"], t.shape)
    sample = torch.from_numpy(data[prefix + "::sample"])
    e_sample = rel_l2(t.reshape(-1)[::stride], sample)
    l2 = float(t.double().norm())
    e_norm = abs(l2 - float(data[prefix + "::l2"])) / (float(data[prefix + "::l2"]) + 1e-30)
    return max(e_sample, e_norm)
