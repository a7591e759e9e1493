"""GPU: the evaluator this build rewrote (stem + 80 tower convolutions on k_conv3x3_c256, BN folded) at FULL depth --
40 blocks x 256 channels, non-trivial BatchNorm statistics, real leaf batches -- against (i) the reference architecture
in float32 and (ii) the path the reference itself runs on a GPU: the same ``Net`` under ``torch.autocast("cuda")``
(reference net.py:178-189). The bar: the fused path is no further from float32 than the reference's own autocast path."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _leaf_batch(B, plies, seed):
    """Real evaluator inputs: B boards after `plies` uniformly random plies each (staggered), as the engine encodes them."""
    from chinesechesszero_amd.engine import SelfPlayEngine
    from chinesechesszero_amd.net import uniform_evaluator
    e = SelfPlayEngine(B, n_playout=1, seed=seed, max_plies=plies + 8)
    temps = np.full(B, 1e3, np.float64)
    idx = np.arange(B)
    for t in range(plies):
        if t:
            e.reset(((idx % plies) == t).astype(np.uint8))
        leaf = e.select_leaves()
        e.expand_backup(*uniform_evaluator(leaf))
        e.finish_move(temps=temps, keep_tree=False)
        if e.game_status()["over"].any():
            e.harvest()
    x = e.select_leaves().clone()
    e.check_healthy()
    e.close()
    return x


def _trained_like_net(dev, calib):
    """Random weights, but BatchNorm statistics that MATCH the activations (one calibration pass in train mode, as training
    would leave them) and random affine parameters: activations stay O(1) through all 40 blocks, like a trained net's."""
    from chinesechesszero_amd.net import Net
    torch.manual_seed(1234)
    net = Net(256, 40).to(dev)
    bns = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    g = torch.Generator(device="cpu").manual_seed(99)
    for m in bns:
        m.momentum = 1.0
        m.weight.data.copy_(torch.empty(m.num_features).uniform_(0.5, 1.5, generator=g))
        m.bias.data.copy_(torch.empty(m.num_features).normal_(0, 0.2, generator=g))
    net.train()
    with torch.no_grad():
        net(calib.float())
    net.eval()
    for m in bns:  # statistics that are close to, not exactly, the batch's (a trained net's running averages)
        m.running_mean.mul_(1.0 + 0.05 * torch.randn(m.num_features, generator=g).to(dev))
        m.running_var.mul_(torch.empty(m.num_features).uniform_(0.8, 1.25, generator=g).to(dev))
    return net


def _metrics(p, v, p_ref, v_ref):
    top1 = float((p.argmax(1) == p_ref.argmax(1)).float().mean())
    t5, t5r = p.topk(5, dim=1).indices, p_ref.topk(5, dim=1).indices
    top5 = float((t5.unsqueeze(2) == t5r.unsqueeze(1)).any(2).float().mean())
    kl = float((p_ref * (torch.log(p_ref.clamp_min(1e-30)) - torch.log(p.clamp_min(1e-30)))).sum(1).mean())
    return {"max_dp": float((p - p_ref).abs().max()), "mean_dp": float((p - p_ref).abs().mean()), "max_dv": float((v - v_ref).abs().max()),
            "mean_dv": float((v - v_ref).abs().mean()), "top1": top1, "top5": top5, "kl": kl}


@pytest.mark.parametrize("B", [4096, 200])
def test_fused_evaluator_at_full_depth_is_no_worse_than_the_reference_autocast_path(B):
    from chinesechesszero_amd.net import InferenceNet
    dev = torch.device("cuda", 0)
    x = _leaf_batch(B, plies=60, seed=3)                       # fp16 [B,17,7,10,9], mixed positions
    net = _trained_like_net(dev, _leaf_batch(512, plies=60, seed=4))
    with torch.no_grad():
        logp32, v32 = net(x.float())                           # (i) reference architecture, float32
        p32, v32 = logp32.exp(), v32.view(-1)
        with torch.autocast("cuda"):                           # (ii) what reference net.py:178-189 runs on a GPU
            logp_ac, v_ac = net(x)
        p_ac, v_ac = logp_ac.float().exp(), v_ac.float().view(-1)
        inf = InferenceNet(net).to(dev).eval()
        assert inf._use_fused_tower(torch.empty((B, 256, 10, 9), dtype=torch.float16, device=dev).contiguous(memory_format=torch.channels_last))
        p16, v16 = inf(x)                                      # this build: stem + tower on k_conv3x3_c256, BN folded
        logits, v16b = inf(x, return_logits=True)
        inf.set_options(fused_conv=False)                      # same folded weights through MIOpen + the one-pass epilogue
        try:
            p_mi, v_mi = inf(x)
        finally:
            inf.set_options(fused_conv=True)
    assert torch.isfinite(p16).all() and torch.isfinite(v16).all() and torch.equal(v16, v16b)
    assert torch.allclose(torch.softmax(logits.float(), 1), p16, atol=1e-6)
    act = float(v32.abs().mean())
    assert 0.02 < act < 0.98 and float(p32.max(1).values.mean()) > 2.0 / 2086   # a net with something to say, not a flat one
    fused, autoc, miopen = _metrics(p16, v16, p32, v32), _metrics(p_ac, v_ac, p32, v32), _metrics(p_mi, v_mi, p32, v32)
    print("evaluator_depth_parity", json.dumps({"B": B, "fused_vs_f32": fused, "autocast_vs_f32": autoc, "miopen_folded_vs_f32": miopen}))
    # the bar: no further from float32 than the reference's own GPU path ...
    assert fused["mean_dp"] <= 1.10 * autoc["mean_dp"] + 1e-9 and fused["mean_dv"] <= 1.10 * autoc["mean_dv"] + 1e-9
    assert fused["max_dp"] <= 1.5 * autoc["max_dp"] + 1e-6 and fused["max_dv"] <= 1.5 * autoc["max_dv"] + 1e-6
    assert fused["kl"] <= 1.10 * autoc["kl"] + 1e-9
    assert fused["top1"] >= autoc["top1"] - 0.01 and fused["top5"] >= autoc["top5"] - 0.01
    # ... and absolute tolerances of fp16 inference at 81 layers
    assert fused["max_dp"] < 2e-2 and fused["max_dv"] < 6e-2 and fused["top1"] > 0.97 and fused["top5"] > 0.97
    # the hand-written kernel and MIOpen compute the same folded network: they differ by summation order only
    assert float((p16 - p_mi).abs().max()) < 1e-2 and float((v16 - v_mi).abs().max()) < 3e-2


def test_shapes_off_the_fused_path_are_logged(caplog):
    """The hand-written kernels serve 256-channel towers at every batch size; a tower of another width takes MIOpen + the
    epilogue pass and says so once per shape."""
    import logging
    from chinesechesszero_amd.net import InferenceNet, Net
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    inf = InferenceNet(Net(128, 1).to(dev).eval()).to(dev).eval()
    x = torch.zeros((64, 17, 7, 10, 9), dtype=torch.float16, device=dev)
    with caplog.at_level(logging.INFO, logger="chinesechesszero_amd"):
        inf(x)
        inf(x)
    msgs = [r.getMessage() for r in caplog.records if "fused" in r.getMessage()]
    assert len(msgs) == 1 and "64 boards x 128 channels" in msgs[0]
    inf256 = InferenceNet(Net(256, 1).to(dev).eval()).to(dev).eval()
    with caplog.at_level(logging.INFO, logger="chinesechesszero_amd"):
        caplog.clear()
        inf256(x)
        inf256(x[:1])
    assert not [r for r in caplog.records if "fused" in r.getMessage()]      # 64 boards and ONE board: both on the fused kernels


def test_a_boards_evaluation_does_not_depend_on_the_batch_size(monkeypatch):
    """One board evaluated alone and in a batch of 24 (k_conv3x3_small), of 90 and of 300 (the 256-pixel tile kernel), of 700 (the
    group-of-16 kernel, from 640 boards on) and -- layout forced -- of 90 and 300 on the group-of-16 kernel (padded to whole
    groups): the stem + tower output of that board is the same, bit for bit -- all three kernels add the same products in the same
    order -- and, since round 4, so are its LOGITS and its VALUE: heads and FC layers are hand-written kernels too
    (csrc/cczero_heads.h), every output element one fixed chain of MFMAs whatever tile it lands in. (Round 3's heads were torch
    GEMMs that pick a kernel per problem size: float16 round-off apart.)"""
    from chinesechesszero_amd.net import InferenceNet, Net
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    inf = InferenceNet(Net(256, 6).to(dev).eval()).to(dev).eval()
    x = _leaf_batch(700, 16, seed=5)
    probe = x[123:124].clone()
    outs = []
    for B, layout in ((1, "auto"), (24, "auto"), (90, "auto"), (300, "auto"), (700, "auto"), (90, "g16"), (300, "g16")):
        inf.set_options(layout=layout)
        assert inf._g16(B) == (B >= 640 or (layout == "g16" and B > 64))
        xb = x[:B].clone()
        xb[B // 2] = probe[0]
        t = inf.tower_activations(xb)
        outs.append(t[B // 2].clone())
        lg, v = inf(xb, return_logits=True)
        if B == 1:
            lg1, v1 = lg[0].clone(), v[0].clone()
        else:
            assert torch.equal(lg[B // 2], lg1) and torch.equal(v[B // 2], v1), (B, layout)
    for t in outs[1:]:
        assert torch.equal(t, outs[0])
    # the torch tail (fused_heads off) agrees to float16 round-off: the kernels compute the reference's layers, not something else
    inf.set_options(layout="auto", fused_heads=False)
    lg_t, v_t = inf(probe, return_logits=True)
    inf.set_options(fused_heads=True)
    assert (lg_t[0].float() - lg1.float()).abs().max().item() < 2e-2 and abs(float(v_t[0] - v1)) < 2e-3


def test_one_board_has_the_same_logits_and_value_at_batch_sizes_1_200_4096_and_on_the_planned_boundary():
    """The full 40 x 256 evaluator: a position alone, among 200 boards (256-pixel tile kernel), among 4096 (group-of-16 kernel, two
    concurrent chains) and as a planned row of a 4096-board batch of which 3000 rows are live: bit-identical logits and value.
    This is what the evaluation cache rests on (a cached evaluation is the evaluation the network would give again, whatever
    batch the position turns up in), now by construction."""
    from chinesechesszero_amd.net import PolicyValueNet
    dev = torch.device("cuda", 0)
    torch.manual_seed(13)
    pvn = PolicyValueNet(device=dev)
    pvn.refresh_inference_copy()
    x = _leaf_batch(4096, 24, seed=17)
    probe = x[2222].clone()
    got = []
    for B, slot in ((1, 0), (200, 137), (4096, 4095), (4096, 16)):
        xb = x[:B].clone()
        xb[slot] = probe
        lg, v = pvn.evaluate_leaves_logits(xb)
        got.append((lg[slot].clone(), v[slot].clone()))
    rows = torch.randperm(4096, device=dev)[:3000].to(torch.int32).contiguous()
    rows[1234] = 2222
    n_rows = torch.tensor([3000], dtype=torch.int32, device=dev)
    lg, v = pvn.evaluate_leaves_logits(x, plan=(rows, n_rows))
    got.append((lg[1234].clone(), v[1234].clone()))
    for lg_i, v_i in got[1:]:
        assert torch.equal(lg_i, got[0][0]) and torch.equal(v_i, got[0][1])
    assert torch.isfinite(got[0][0].float()).all() and abs(float(got[0][1])) < 1


def test_a_boards_evaluation_does_not_depend_on_its_slot_in_the_batch():
    """The WHOLE evaluator (pack + stem + 80 tower convolutions + heads GEMM + FC layers), 40 x 256 at 4096 rows: the same
    position placed in different rows of the batch -- first and last row of a pixel tile, either side of the 2048-board group
    and of the 1024-board chain boundaries, the last row -- and among different neighbours gets bit-identical logits and value.
    (What "results do not depend on which rank / slot a game runs in" rests on, at equal batch size.)"""
    from chinesechesszero_amd.net import PolicyValueNet
    dev = torch.device("cuda", 0)
    torch.manual_seed(7)
    pvn = PolicyValueNet(device=dev)
    pvn.refresh_inference_copy()
    B = 4096
    x = _leaf_batch(B, 24, seed=11)
    probe = x[777].clone()
    slots = [0, 1, 2, 255, 256, 1023, 1024, 2047, 2048, 3071, 3072, 4094, 4095]
    xa = x.clone()
    xa[slots] = probe
    la, va = pvn.evaluate_leaves_logits(xa)
    la, va = la.clone(), va.clone()
    for s in slots[1:]:
        assert torch.equal(la[s], la[slots[0]]) and torch.equal(va[s], va[slots[0]]), s
    # other neighbours (the batch reversed around the probe rows): the probe's numbers do not move
    xb = x.flip(0).contiguous()
    xb[slots] = probe
    lb, vb = pvn.evaluate_leaves_logits(xb)
    for s in slots:
        assert torch.equal(lb[s], la[slots[0]]) and torch.equal(vb[s], va[slots[0]]), s
    # and every untouched row of the reversed batch equals its mirror row of the first batch
    keep = torch.ones(B, dtype=torch.bool, device=dev)
    keep[slots] = False
    keep &= keep.flip(0)
    assert torch.equal(lb[keep], la.flip(0)[keep]) and torch.equal(vb[keep], va.flip(0)[keep])


def test_both_row_layouts_give_the_same_evaluation(monkeypatch):
    """The evaluator at full depth (40 x 256) and 4096 rows in the group-of-16 row layout (k_conv3x3_g16: whole-rank tiles, taps off
    the board skipped) and in board-major rows (k_conv3x3_c256, `CCZ_CONV_LAYOUT=nhwc`): the same logits and values, bit for bit --
    the kernels add the same products in the same order, and a skipped product of zeros cannot change a sum. Also on the planned
    (compacted) boundary, and -- tower activations, the heads' GEMMs see another row count there -- with a batch that the
    group-of-16 layout pads to whole groups (3990 boards)."""
    from chinesechesszero_amd.net import PolicyValueNet
    dev = torch.device("cuda", 0)
    torch.manual_seed(9)
    pvn = PolicyValueNet(device=dev)
    pvn.refresh_inference_copy()
    x = _leaf_batch(4096, 24, seed=21)
    rows = torch.randperm(4096, device=dev)[:3000].to(torch.int32).contiguous()
    n_rows = torch.tensor([3000], dtype=torch.int32, device=dev)
    out = {}
    # "+edge": the middle + edge-pair launches of the group-of-16 kernel; "-sep": the head convolutions as a pass of their own over
    # the stored output of the tower instead of in the last layer's epilogue (round 4; the default fuses them on group-of-16 rows)
    variants = ("g16", "nhwc", "g16+edge", "g16-sep", "g16+edge-sep")
    for layout in variants:
        pvn._infer.set_options(layout=layout[:3] if layout.startswith("g16") else layout, edge_tiles="+edge" in layout, fused_last="-sep" not in layout)
        full = pvn.evaluate_leaves_logits(x)
        part = pvn._infer.tower_activations(x[:3990].contiguous())
        plan = pvn.evaluate_leaves_logits(x, plan=(rows, n_rows))
        odd = pvn.evaluate_leaves_logits(x[:3976].contiguous())     # 249 groups (the edge kernel pairs them: one left over), the last one 8 boards
        out[layout] = [t.clone() for t in (*full, part, plan[0][:3000], plan[1][:3000], *odd)]
    for v in variants[1:]:
        for a, b in zip(out["g16"], out[v]):
            assert torch.equal(a, b), v
    pvn._infer.set_options(layout="auto", edge_tiles="auto", fused_last=True)
    assert torch.equal(out["g16"][5], out["g16"][0][:3976]) and torch.equal(out["g16"][6], out["g16"][1][:3976])
    # the planned rows are the rows of the full batch
    assert torch.equal(out["g16"][3], out["g16"][0][rows.long()]) and torch.equal(out["g16"][4], out["g16"][1][rows.long()])
    assert out["g16"][2].shape == (3990, 256, 10, 9)
