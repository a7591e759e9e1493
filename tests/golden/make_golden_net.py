#!/usr/bin/env python3
"""Generate tests/golden/reference_net.{npz,json} by EXECUTING the reference's own net.py (build container only).

``net.py`` imports the absent ``cchess`` module for two names on this path (``cchess.Move.uci``, ``cchess.RED``): an empty
placeholder module carrying them is registered, exactly as make_golden.py does for tools.py / mcts.py. Pinned here:
  * the ``state_dict`` keys and shapes of the reference ``Net`` (what a reference-trained .pkl holds),
  * ``Net.forward`` (net.py:82-110) at the reference's full size 40 x 256 on fixed inputs, float32 on the CPU, with weights
    given by the closed-form recipe of net_recipe.py (nothing but outputs is stored),
  * ``PolicyValueNet.policy_value_fn`` (net.py:151-205) end to end on two positions: legal-move gathering, the
    [1,17,7,10,9] input it builds (zeros in 7 of 8 history slots, current position in slot 7 / 15, turn plane), exp(log p).
Outputs are data only; no reference source text is stored.
"""
import json
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("CCZ_GOLDEN_OUT", HERE)   # where the fixtures are written (tests/test_cpu_golden_regenerates.py: a scratch directory)
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from _ref_guard import assert_reference_untouched, silence_reference_log  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))

import net_recipe  # noqa: E402
from oracle import OracleBoard  # noqa: E402

REF = "/root/reference"


def load_reference():
    ph = types.ModuleType("cchess")
    ph.RED, ph.BLACK = True, False

    class Move:
        @staticmethod
        def from_uci(s):
            return s

        @staticmethod
        def uci(m):  # net.py:155-156 calls cchess.Move.uci(move) on the items of board.legal_moves
            return m

    ph.Move = Move
    sys.modules["cchess"] = ph
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir("/tmp")
    import net as ref_net  # noqa
    os.chdir(cwd)
    silence_reference_log(ref_net)   # tools.log writes next to tools.py whatever the working directory is (tools.py:46-51)
    return ref_net


def main():
    torch.set_num_threads(8)
    ref_net = load_reference()
    out, meta = {}, {}
    net = ref_net.Net().eval()                       # the reference's own architecture at its default size
    meta["state_dict"] = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in net.state_dict().items()]
    meta["n_params"] = int(sum(p.numel() for p in net.parameters()))
    net_recipe.fill_state_dict(net)
    x = torch.from_numpy(net_recipe.inputs(3))
    with torch.no_grad():
        logp, v = net(x)
    out["forward_logp"] = logp.numpy()
    out["forward_value"] = v.numpy()
    # policy_value_fn on the CPU path (use_gpu=False): the evaluator boundary as mcts.py:114 calls it
    pvn = ref_net.PolicyValueNet(use_gpu=False)
    net_recipe.fill_state_dict(pvn.policy_value_net)
    from golden_cases import STARTS
    boards = {"start": OracleBoard(), "wide80_black": OracleBoard.from_array(STARTS["wide80"], 0, 0)}
    for name, b in boards.items():
        act_probs, value = pvn.policy_value_fn(b)
        pairs = list(act_probs)
        out[f"pvfn_{name}_ids"] = np.array([a for a, _ in pairs], dtype=np.int32)
        out[f"pvfn_{name}_probs"] = np.array([p for _, p in pairs], dtype=np.float32)
        out[f"pvfn_{name}_value"] = np.asarray(value, dtype=np.float32)
        out[f"pvfn_{name}_sq"] = b.squares()
        out[f"pvfn_{name}_turn"] = np.int32(1 if b.turn else 0)
        meta[f"pvfn_{name}_value_shape"] = list(np.asarray(value).shape)
        meta[f"pvfn_{name}_value_dtype"] = str(np.asarray(value).dtype)
    # ---- PolicyValueNet.train_step (net.py:212-247) and policy_value (net.py:137-148): one Adam step on a fixed batch
    # from the recipe weights (train-mode BatchNorm on the batch, l2 inside the optimiser), then the batched evaluation
    net_recipe.fill_state_dict(pvn.policy_value_net)
    xb = net_recipe.inputs(4)
    pib = np.abs(net_recipe._wave("train_pi", 4 * 2086).reshape(4, 2086)) ** 4
    pib = (pib / pib.sum(1, keepdims=True)).astype(np.float32)
    zb = np.array([1.0, -1.0, 0.0, 1.0], dtype=np.float32)
    act_probs, value = pvn.policy_value(xb)            # batched evaluation with the recipe weights (eval mode)
    out["policy_value_probs"] = np.asarray(act_probs)
    out["policy_value_value"] = np.asarray(value)
    loss, entropy = pvn.train_step(xb, pib, zb, lr=0.002)
    out["train_loss"] = np.asarray(loss, dtype=np.float64)
    out["train_entropy"] = np.asarray(entropy, dtype=np.float64)
    sd = pvn.policy_value_net.state_dict()
    for k in ("conv_block.weight", "res_blocks.39.conv2.weight", "policy_fc.bias", "value_fc2.weight", "conv_block_bn.running_mean"):
        out["train_after_" + k] = sd[k].detach().numpy().ravel()[:16].copy()
    out["train_pi"] = pib

    np.savez_compressed(os.path.join(OUT, "reference_net.npz"), **out)
    with open(os.path.join(OUT, "reference_net.json"), "w") as f:
        json.dump(meta, f, indent=1)
    p = np.exp(out["forward_logp"])
    print("params", meta["n_params"], "max p", p.max(1), "entropy", -(p * out["forward_logp"]).sum(1), "value", out["forward_value"].ravel())
    print({k: out[k] for k in out if k.endswith("_value")})


if __name__ == "__main__":
    main()
    assert_reference_untouched()
