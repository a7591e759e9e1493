"""Deterministic weights and inputs for the policy-value net goldens (this build's own code, shared by the generator that
runs the REFERENCE net.py and by the tests that run this build's net.py): every tensor of a ``state_dict`` is a closed-form
function of its name and shape, so no weight file has to be stored to give both sides identical parameters."""
import zlib

import numpy as np


def _wave(name: str, n: int) -> np.ndarray:
    seed = zlib.crc32(name.encode()) % 9973
    i = np.arange(n, dtype=np.float64)
    return np.sin(i * 0.6180339887498949 + seed * 0.37) * np.cos(i * 0.0137 + seed)


def tensor_for(name: str, shape) -> np.ndarray:
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if shape else 1
    x = _wave(name, n)
    if name.endswith("num_batches_tracked"):
        return np.zeros(shape, dtype=np.int64)
    if name.endswith("running_var"):
        v = 1.0 + 0.3 * x
    elif name.endswith("running_mean"):
        v = 0.1 * x
    elif "_bn" in name and name.endswith("weight"):
        v = 1.0 + 0.2 * x
    elif "_bn" in name and name.endswith("bias"):
        v = 0.05 * x
    elif name.endswith("bias"):
        v = 0.05 * x
    else:  # conv / linear weights: fan-in scaled; the second conv of a block smaller, so that 40 residual blocks stay O(1)
        fan_in = int(np.prod(shape[1:]))
        gain = 0.9 if ".conv2." in name else (5.0 if "_fc" in name else 2.0)   # sharp heads: a flat policy would hide flatten-order errors
        v = gain * x / np.sqrt(fan_in)
    return v.reshape(shape).astype(np.float32)


def fill_state_dict(module) -> None:
    import torch
    sd = module.state_dict()
    new = {k: torch.from_numpy(tensor_for(k, v.shape)).to(v.dtype) for k, v in sd.items()}
    module.load_state_dict(new)


def inputs(n: int = 3) -> np.ndarray:
    """n evaluator inputs [n,17,7,10,9] (0/1 planes incl. history groups, as the training path feeds the net)."""
    x = (_wave("inputs", n * 17 * 7 * 90).reshape(n, 17, 7, 10, 9) > 0.55).astype(np.float32)
    x[:, 16] = (np.arange(n) % 2).reshape(n, 1, 1, 1)
    return x
