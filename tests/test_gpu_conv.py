"""The evaluator's tower convolution kernel (ccz_conv3x3_c256_f16, hand-written MFMA implicit GEMM) against a
float32 torch convolution of the same fp16-rounded operands, and the fused tower against the reference
architecture (net.py:20-43, 53-110). Tolerances: fp32 accumulation, one fp16 rounding of the result (two with a residual)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _conv(x, w, bias32, res, y, relu):
    from chinesechesszero_amd import _lib
    s = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(_lib.lib().ccz_conv3x3_c256_f16(s, C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(bias32.data_ptr()),
                                               C.c_void_p(res.data_ptr()) if res is not None else None, C.c_void_p(y.data_ptr()),
                                               x.shape[0] * 90, relu))
    return y


@pytest.mark.parametrize("boards", [1, 2, 3, 17, 64, 257])
def test_conv_kernel_matches_fp32_convolution(boards):
    """Tile edges fall inside boards (256 pixels per tile, 90 per board); the last tile is partial for every size here."""
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(boards)
    cl = torch.channels_last
    x = (torch.randn(boards, 256, 10, 9, generator=g) * 0.7).to(dev).half().contiguous(memory_format=cl)
    r = (torch.randn(boards, 256, 10, 9, generator=g) * 0.7).to(dev).half().contiguous(memory_format=cl)
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.03).to(dev).half().contiguous(memory_format=cl)
    b = (torch.randn(256, generator=g) * 0.2).to(dev)
    ref = F.conv2d(x.float(), w.float(), b, padding=1)
    for res, relu in ((None, 1), (r, 1), (None, 0), (r, 0)):
        for force_tile in (0, 32):   # the kernel the batch size selects (small batches: k_conv3x3_small), and the tile kernel
            y = torch.full_like(x, float("nan"))
            _conv(x, w, b, res, y, relu | force_tile)
            want = ref if res is None else ref + res.float()
            if relu:
                want = F.relu(want)
            err = (y.float() - want).abs().max().item()
            assert err < 4e-3 * max(1.0, want.abs().max().item()), (boards, res is not None, relu, force_tile, err)
    # flag bit 1 (descending tile order) changes nothing in the result
    ya, yb = _conv(x, w, b, r, torch.empty_like(x), 1 | 32), _conv(x, w, b, r, torch.empty_like(x), 3 | 32)
    assert torch.equal(ya, yb)
    # the output may be written over the residual input (how the tower uses it)
    y = r.clone(memory_format=torch.preserve_format)
    _conv(x, w, b, y, y, 1)
    want = F.relu(ref + r.float())
    assert (y.float() - want).abs().max().item() < 4e-3 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("boards", [1, 2, 3, 7, 24, 25, 64, 65, 130])
def test_small_batch_kernel_is_bit_identical_to_the_tile_kernel(boards):
    """k_conv3x3_small (a 16-channel x 64-pixel block per workgroup: what batches of up to 64 boards run on) performs the same operations
    in the same order as k_conv3x3_c256: identical bits, with and without residual / ReLU, tower and stem shape; both against
    float32 as well. Flag bit 4 forces the small kernel, bit 5 the tile kernel."""
    from chinesechesszero_amd import _lib
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(100 + boards)
    cl = torch.channels_last
    x = (torch.randn(boards, 256, 10, 9, generator=g) * 0.7).to(dev).half().contiguous(memory_format=cl)
    r = (torch.randn(boards, 256, 10, 9, generator=g) * 0.7).to(dev).half().contiguous(memory_format=cl)
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.03).to(dev).half().contiguous(memory_format=cl)
    b = (torch.randn(256, generator=g) * 0.2).to(dev)
    ref = F.conv2d(x.float(), w.float(), b, padding=1)
    for res, relu in ((None, 1), (r, 1), (r, 0)):
        ys = _conv(x, w, b, res, torch.full_like(x, float("nan")), relu | 16)
        yt = _conv(x, w, b, res, torch.full_like(x, float("nan")), relu | 32)
        assert torch.equal(ys, yt), (boards, res is not None, relu)
        want = ref if res is None else ref + res.float()
        if relu:
            want = F.relu(want)
        assert (ys.float() - want).abs().max().item() < 4e-3 * max(1.0, want.abs().max().item())
    # in place over the residual, as the tower calls it
    ya, yb = r.clone(memory_format=torch.preserve_format), r.clone(memory_format=torch.preserve_format)
    _conv(x, w, b, ya, ya, 1 | 16)
    _conv(x, w, b, yb, yb, 1 | 32)
    assert torch.equal(ya, yb)
    # the stem shape (one 64-channel chunk)
    L = _lib.lib()
    s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    x64 = torch.zeros(boards * 90, 64, device=dev, dtype=torch.float16)
    x64[:, :21] = (torch.rand(boards * 90, 21, generator=g) > 0.8).to(dev).half()
    w64 = (torch.randn(256, 3, 3, 64, generator=g) * 0.05).to(dev).half()
    outs = []
    for flag in (16, 32):
        y = torch.full((boards * 90, 256), float("nan"), device=dev, dtype=torch.float16)
        _lib.check(L.ccz_conv3x3_stem_f16(s, C.c_void_p(x64.data_ptr()), C.c_void_p(w64.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(y.data_ptr()), boards * 90, 1 | flag))
        outs.append(y)
    assert torch.equal(outs[0], outs[1]) and not torch.isnan(outs[0]).any()


def test_conv_kernel_board_edges_are_zero_padded_per_board():
    """A single hot pixel in a corner must not leak into the neighbouring board or wrap around a file edge:
    with all-ones weights the output is the 3x3 box count of the hot pixels of THAT board."""
    dev = torch.device("cuda", 0)
    cl = torch.channels_last
    x = torch.zeros(4, 256, 10, 9, device=dev, dtype=torch.float16).contiguous(memory_format=cl)
    x[0, 0, 9, 8] = 1   # last pixel of board 0
    x[1, 0, 0, 0] = 1   # first pixel of board 1
    x[2, 0, 4, 8] = 1   # right edge, middle rank
    x[3, 0, 5, 0] = 1   # left edge
    w = torch.zeros(256, 256, 3, 3, device=dev, dtype=torch.float16).contiguous(memory_format=cl)
    w[:, 0] = 1
    b = torch.zeros(256, device=dev)
    y = _conv(x, w, b, None, torch.empty_like(x), 0)
    want = F.conv2d(x.float(), w.float(), None, padding=1)
    assert torch.equal(y.float(), want)
    assert y[0, 5].sum().item() == 4 and y[1, 5].sum().item() == 4 and y[2, 5].sum().item() == 6


def test_conv_kernel_rejects_bad_arguments():
    from chinesechesszero_amd import _lib
    dev = torch.device("cuda", 0)
    x = torch.zeros(90 * 256, device=dev, dtype=torch.float16)
    b = torch.zeros(256, device=dev)
    L = _lib.lib()
    s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    p = C.c_void_p(x.data_ptr())
    assert L.ccz_conv3x3_c256_f16(s, p, p, C.c_void_p(b.data_ptr()), None, p, 90, 1) < 0          # y aliases x
    assert L.ccz_conv3x3_c256_f16(s, p, p, C.c_void_p(b.data_ptr()), None, C.c_void_p(x.data_ptr() + 16), 89, 1) < 0  # not boards * 90
    assert L.ccz_conv3x3_c256_f16(s, None, p, C.c_void_p(b.data_ptr()), None, p, 90, 1) < 0


def test_fused_tower_matches_reference_architecture_and_miopen_path(monkeypatch):
    """256-channel tower (the only width the kernel serves), 3 blocks, 200 boards: fused path vs fp32 reference
    architecture, and vs the MIOpen + epilogue path it replaces."""
    from chinesechesszero_amd.net import InferenceNet, Net
    dev = torch.device("cuda", 0)
    torch.manual_seed(5)
    net = Net(256, 3).to(dev).eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 2)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
    B = 200
    x = torch.zeros(B, 17, 7, 10, 9, device=dev)
    x[:, 7] = (torch.rand(B, 7, 10, 9, device=dev) > 0.9).float()
    x[:, 15] = (torch.rand(B, 7, 10, 9, device=dev) > 0.9).float()
    x[::2, 16] = 1
    with torch.no_grad():
        logp, v = net(x)
        inf = InferenceNet(net).to(dev).eval()
        assert inf._use_fused_tower(torch.empty(B, 256, 10, 9, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last))
        p_f, v_f = inf(x.half())
        inf.set_options(fused_conv=False)
        p_m, v_m = inf(x.half())
        inf.set_options(fused_conv=True)
        lg_f, _ = inf(x.half(), return_logits=True)
    assert (logp.exp() - p_f).abs().max().item() < 2e-3 and (v.view(-1) - v_f).abs().max().item() < 2e-2
    assert (p_m - p_f).abs().max().item() < 2e-3 and (v_m - v_f).abs().max().item() < 2e-2
    assert torch.allclose(p_f.sum(1), torch.ones(B, device=dev), atol=1e-3)
    assert lg_f.shape == (B, 2086)
    # small batches run on the hand-written kernels too (k_conv3x3_small; round 2 sent them to MIOpen)
    assert inf._use_fused_tower(torch.empty(8, 256, 10, 9, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last))


def test_fused_tower_board_ranges_on_several_streams_equal_one_chain(monkeypatch):
    """The tower cuts the batch into independent board ranges on separate HIP streams (net.py _tower_fused): same bits as
    one chain, whatever the cut (a board's arithmetic does not depend on which tile it falls in)."""
    from chinesechesszero_amd.net import InferenceNet, Net
    dev = torch.device("cuda", 0)
    torch.manual_seed(6)
    net = Net(256, 2).to(dev).eval()
    inf = InferenceNet(net).to(dev).eval()
    for B in (600, 608):   # 600 boards: board-major rows, 256-pixel tiles; 608 = 38 groups of 16: the group-of-16 layout (forced)
        inf.set_options(layout="g16" if B == 608 else "auto")
        x0 = torch.relu(torch.randn(B, 256, 10, 9, device=dev)).half().contiguous(memory_format=torch.channels_last)
        outs = []
        for chains in (1, 8, 3):
            inf.TOWER_CHAINS = chains
            inf._chain_streams = None
            outs.append(inf._tower_fused(x0.clone(memory_format=torch.preserve_format)).clone())
            torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        assert torch.isfinite(outs[0].float()).all() and outs[0].abs().max().item() > 0


def test_heads_in_the_last_layer_under_every_launch_structure():
    """The whole evaluator (2 blocks x 256, group-of-16 rows) with the head convolutions inside the last tower layer: logits and
    values are the same bits whatever the launch structure -- 1 / 3 chains, 1 / 2 sequential board groups, edge-pair tiles or not,
    a batch that pads its last group, whole batch or planned rows -- and the same as with the heads as a pass of their own."""
    from chinesechesszero_amd.net import InferenceNet, Net
    dev = torch.device("cuda", 0)
    torch.manual_seed(8)
    net = Net(256, 2).to(dev).eval()
    inf = InferenceNet(net).to(dev).eval()
    g = torch.Generator().manual_seed(5)
    for B in (1300, 1296):   # 82 groups, the last one holds 4 boards; 81 whole groups (odd: the edge kernel pairs one with itself)
        leaf = torch.zeros(B, 17, 7, 10, 9, dtype=torch.float16)
        leaf.view(B, 119, 90)[:, 49:56] = (torch.rand(B, 7, 90, generator=g) < 0.1).half()
        leaf.view(B, 119, 90)[:, 105:119] = (torch.rand(B, 14, 90, generator=g) < 0.1).half()
        leaf = leaf.to(dev)
        rows = torch.randperm(B, generator=g)[:B - 200].to(torch.int32).to(dev).contiguous()
        n_rows = torch.tensor([B - 200], dtype=torch.int32, device=dev)
        want = None
        for fused_last in (False, True):
            for chains, groups, edge in ((1, 1, False), (3, 1, True), (2, 2, False), (3, 2, True)):
                inf.set_options(layout="g16", fused_last=fused_last, chains=chains, groups=groups, edge_tiles=edge)
                inf._chain_streams = None
                full = inf(leaf, return_logits=True)
                plan = inf(leaf, return_logits=True, plan=(rows, n_rows))
                got = [full[0].clone(), full[1].clone(), plan[0][:B - 200].clone(), plan[1][:B - 200].clone()]
                torch.cuda.synchronize()
                if want is None:
                    want = got
                    assert torch.isfinite(got[0].float()).all() and float(got[0].float().abs().max()) > 0
                    assert torch.equal(got[2], got[0][rows.long()]) and torch.equal(got[3], got[1][rows.long()])
                else:
                    for a, b in zip(want, got):
                        assert torch.equal(a, b), (B, fused_last, chains, groups, edge)


def test_conv_kernel_full_size_board_permutation_and_sample():
    """BASELINE size (4096 boards = 1440 tiles): (1) boards are independent, so permuting the boards of the input permutes
    the output bit for bit, whatever tile / wave / lane a board lands in; (2) a sample of boards against float32."""
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(11)
    cl = torch.channels_last
    B = 4096
    x = torch.relu(torch.randn(B, 256, 10, 9, generator=g) * 0.7).to(dev).half().contiguous(memory_format=cl)
    r = (torch.randn(B, 256, 10, 9, generator=g) * 0.7).to(dev).half().contiguous(memory_format=cl)
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.03).to(dev).half().contiguous(memory_format=cl)
    b = (torch.randn(256, generator=g) * 0.2).to(dev)
    y = _conv(x, w, b, r, torch.empty_like(x), 1)
    perm = torch.randperm(B, generator=g).to(dev)
    xp = x[perm].contiguous(memory_format=cl)
    rp = r[perm].contiguous(memory_format=cl)
    yp = _conv(xp, w, b, rp, torch.empty_like(x), 1)
    assert torch.equal(yp, y[perm])
    sample = [0, 1, 2, 1023, 2047, 2048, 4094, 4095]
    want = F.relu(F.conv2d(x[sample].float(), w.float(), b, padding=1) + r[sample].float())
    assert (y[sample].float() - want).abs().max().item() < 4e-3 * max(1.0, want.abs().max().item())
    assert torch.isfinite(y.float()).all()


@pytest.mark.parametrize("boards", [1, 65, 300])
def test_stem_pack_and_convolution_match_reference_ops(boards):
    """ccz_pack_live_planes_f16 (bit-exact gather of planes 49..55, 105..118 into 64-channel NHWC rows) and
    ccz_conv3x3_stem_f16 (one 64-channel chunk of the tower kernel) against torch ops on the same operands."""
    from chinesechesszero_amd import _lib
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(100 + boards)
    leaf = (torch.rand(boards, 17, 7, 10, 9, generator=g) > 0.8).to(dev).half()
    leaf[:, :7] = 1  # planes outside the live set must not leak in
    L = _lib.lib()
    s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    x64 = torch.full((boards, 90, 64), float("nan"), device=dev, dtype=torch.float16)
    _lib.check(L.ccz_pack_live_planes_f16(s, C.c_void_p(leaf.data_ptr()), C.c_void_p(x64.data_ptr()), boards))
    planes = leaf.view(boards, 119, 90)
    want = torch.zeros(boards, 90, 64, device=dev, dtype=torch.float16)
    want[..., :7] = planes[:, 49:56].permute(0, 2, 1)
    want[..., 7:21] = planes[:, 105:119].permute(0, 2, 1)
    assert torch.equal(x64, want)
    w21 = (torch.randn(256, 21, 3, 3, generator=g) * 0.1).to(dev).half()
    b = (torch.randn(256, generator=g) * 0.2).to(dev)
    w64 = torch.zeros(256, 3, 3, 64, device=dev, dtype=torch.float16)
    w64[..., :21] = w21.permute(0, 2, 3, 1)
    y = torch.full((boards, 256, 10, 9), float("nan"), device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    _lib.check(L.ccz_conv3x3_stem_f16(s, C.c_void_p(x64.data_ptr()), C.c_void_p(w64.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(y.data_ptr()), boards * 90, 1))
    x21 = torch.cat([planes[:, 49:56], planes[:, 105:119]], 1).view(boards, 21, 10, 9).float()
    ref = F.relu(F.conv2d(x21, w21.float(), b, padding=1))
    assert (y.float() - ref).abs().max().item() < 4e-3 * max(1.0, ref.abs().max().item())


def test_fused_evaluator_inside_hipgraph_capture_equals_eager():
    """Inside a stream capture the tower uses one chain on the capturing stream (no side streams); the replayed graph
    must give the same tower output bit for bit as the eager two-chain run (the PyTorch head GEMMs may pick other
    kernels under capture: the final probabilities / values are compared at fp16-inference tolerance)."""
    from chinesechesszero_amd.net import InferenceNet, Net
    dev = torch.device("cuda", 0)
    torch.manual_seed(8)
    inf = InferenceNet(Net(256, 2).to(dev).eval()).to(dev).eval()
    B = 512
    leaf = (torch.rand(B, 17, 7, 10, 9, device=dev) > 0.9).half()

    def body(x):
        t = inf.tower_activations(x)
        p, v = inf(x)
        return t, p, v
    with torch.no_grad():
        t_eager, p_eager, v_eager = body(leaf)
        t_eager = t_eager.clone()
        static_in = leaf.clone()
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(2):
                body(static_in)
        torch.cuda.current_stream(dev).wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            t_graph, p_graph, v_graph = body(static_in)
        g.replay()
        torch.cuda.synchronize()
    assert torch.equal(t_graph, t_eager)
    assert torch.allclose(p_graph, p_eager, atol=2e-3) and torch.allclose(v_graph, v_eager, atol=2e-2)


def _to_g16(t):
    """[B, C, 10, 9] channels-last (rows b * 90 + pos) -> the same bytes reordered to rows (g * 90 + pos) * 16 + j, board 16 g + j"""
    B, Cn = t.shape[0], t.shape[1]
    return t.permute(0, 2, 3, 1).reshape(B // 16, 16, 90, Cn).permute(0, 2, 1, 3).contiguous()


def _pack_w(w_nhwc, cin):
    """ccz_pack_conv_weights_g16_f16 on [256, 3, 3, cin] weights; must equal the torch twin the evaluator uses"""
    from chinesechesszero_amd import _lib
    from chinesechesszero_amd.net import pack_conv_weights_g16
    wp = torch.empty(cin // 32, 9, 256, 32, dtype=torch.float16, device=w_nhwc.device)
    s = C.c_void_p(torch.cuda.current_stream(w_nhwc.device).cuda_stream)
    _lib.check(_lib.lib().ccz_pack_conv_weights_g16_f16(s, C.c_void_p(w_nhwc.data_ptr()), C.c_void_p(wp.data_ptr()), cin))
    assert torch.equal(wp, pack_conv_weights_g16(w_nhwc.view(256, 3, 3, cin)))
    return wp


def _from_g16(t, B):
    Cn = t.shape[-1]
    return t.reshape(B // 16, 90, 16, Cn).permute(0, 2, 1, 3).reshape(B, 10, 9, Cn).permute(0, 3, 1, 2)


@pytest.mark.parametrize("boards", [16, 80, 96, 272])
def test_group_of_16_kernel_gives_the_tile_kernels_values(boards):
    """k_conv3x3_g16 (rows in the CCZ_CONV_G16 layout: tiles of two whole ranks of 16 boards, the taps that leave the board
    skipped instead of multiplied by zero rows) adds the same products in the same order as the 256-pixel tile kernel and
    k_conv3x3_small: equal values (a skipped product of zeros can only change the sign of a zero), tower and stem shape, with and
    without residual / ReLU, either tile order, output over the residual; and float32 as the common reference."""
    from chinesechesszero_amd import _lib
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(500 + boards)
    cl = torch.channels_last
    L = _lib.lib()
    s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    x = torch.relu(torch.randn(boards, 256, 10, 9, generator=g) * 0.7).to(dev).half().contiguous(memory_format=cl)
    r = (torch.randn(boards, 256, 10, 9, generator=g) * 0.7).to(dev).half().contiguous(memory_format=cl)
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.03).to(dev).half().contiguous(memory_format=cl)
    b = (torch.randn(256, generator=g) * 0.2).to(dev)
    xg, rg = _to_g16(x), _to_g16(r)
    wp = _pack_w(w.permute(0, 2, 3, 1), 256)     # the group-of-16 form takes its weights packed (contiguous half-tiles)
    ref = F.conv2d(x.float(), w.float(), b, padding=1)
    for res, relu in ((None, 1), (r, 1), (None, 0), (r, 3)):
        y_tile = _conv(x, w, b, res, torch.full_like(x, float("nan")), (relu & 1) | _lib.CONV_FORCE_TILE)
        yg = torch.full_like(xg, float("nan"))
        _lib.check(L.ccz_conv3x3_c256_f16(s, C.c_void_p(xg.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(b.data_ptr()),
                                          C.c_void_p(rg.data_ptr()) if res is not None else None, C.c_void_p(yg.data_ptr()), boards * 90,
                                          relu | _lib.CONV_G16))
        got = _from_g16(yg, boards)
        assert torch.equal(got, y_tile), (boards, res is not None, relu, (got.float() - y_tile.float()).abs().max().item())
        # round 4's middle launch (ranks 1..8) + edge-pair launch (ranks 0 / 9 of two groups per tile, six live taps) against the
        # single launch of five tiles per group (one group: a single launch either way; 5 and 17 groups: the last pair is one group twice)
        y1 = torch.full_like(xg, float("nan"))
        _lib.check(L.ccz_conv3x3_c256_f16(s, C.c_void_p(xg.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(b.data_ptr()),
                                          C.c_void_p(rg.data_ptr()) if res is not None else None, C.c_void_p(y1.data_ptr()), boards * 90,
                                          relu | _lib.CONV_G16 | _lib.CONV_G16_EDGE_TILES))
        assert torch.equal(y1, yg)
        want = ref if res is None else ref + r.float()
        if relu & 1:
            want = F.relu(want)
        assert (got.float() - want).abs().max().item() < 4e-3 * max(1.0, want.abs().max().item())
    # output written over the residual (how the tower uses it)
    yg = rg.clone()
    _lib.check(L.ccz_conv3x3_c256_f16(s, C.c_void_p(xg.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(yg.data_ptr()),
                                      C.c_void_p(yg.data_ptr()), boards * 90, 1 | _lib.CONV_G16))
    assert torch.equal(_from_g16(yg, boards), _conv(x, w, b, r, torch.empty_like(x), 1 | _lib.CONV_FORCE_TILE))
    # stem shape (64 input channels = two chunks of 32) through the layout-aware pack
    leaf = (torch.rand(boards, 17, 7, 10, 9, generator=g) > 0.85).to(dev).half()
    w64 = torch.zeros(256, 3, 3, 64, dtype=torch.float16, device=dev)
    w64[..., :21] = (torch.randn(256, 3, 3, 21, generator=g) * 0.1).to(dev).half()
    x64, x64g = torch.empty(boards * 90, 64, dtype=torch.float16, device=dev), torch.full((boards * 90, 64), float("nan"), dtype=torch.float16, device=dev)
    _lib.check(L.ccz_pack_live_planes_f16(s, C.c_void_p(leaf.data_ptr()), C.c_void_p(x64.data_ptr()), boards))
    _lib.check(L.ccz_pack_live_planes_g16_f16(s, C.c_void_p(leaf.data_ptr()), C.c_void_p(x64g.data_ptr()), boards, None, None))
    assert torch.equal(x64g.view(boards // 16, 90, 16, 64).permute(0, 2, 1, 3).reshape(boards * 90, 64), x64)
    ys = torch.empty(boards * 90, 256, dtype=torch.float16, device=dev)
    ysg = torch.empty_like(ys)
    _lib.check(L.ccz_conv3x3_stem_f16(s, C.c_void_p(x64.data_ptr()), C.c_void_p(w64.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(ys.data_ptr()),
                                      boards * 90, 1 | _lib.CONV_FORCE_TILE))
    _lib.check(L.ccz_conv3x3_stem_f16(s, C.c_void_p(x64g.data_ptr()), C.c_void_p(_pack_w(w64, 64).data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(ysg.data_ptr()),
                                      boards * 90, 1 | _lib.CONV_G16))
    assert torch.equal(ysg.view(boards // 16, 90, 16, 256).permute(0, 2, 1, 3).reshape(boards * 90, 256), ys)
    ysg2 = torch.full_like(ysg, float("nan"))   # the stem shape through the middle + edge-pair launches as well
    _lib.check(L.ccz_conv3x3_stem_f16(s, C.c_void_p(x64g.data_ptr()), C.c_void_p(_pack_w(w64, 64).data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(ysg2.data_ptr()),
                                      boards * 90, 1 | _lib.CONV_G16 | _lib.CONV_G16_EDGE_TILES))
    assert torch.equal(ysg2, ysg)
    # not a multiple of 16 boards: refused
    assert L.ccz_conv3x3_c256_f16(s, C.c_void_p(xg.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(b.data_ptr()), None, C.c_void_p(yg.data_ptr()),
                                  (boards - 1) * 90, 1 | _lib.CONV_G16) != 0


@pytest.mark.parametrize("boards,nwg", [(16, 0), (80, 8), (96, 16), (272, 24), (272, 40), (1024, 0), (1040, 248)])
def test_persistent_group_of_16_kernel_is_bit_identical(boards, nwg):
    """k_conv3x3_g16_pers (CCZ_CONV_G16_PERSISTENT: a fixed number of workgroups walking tile lists, the next tile's slab staged during
    the last chunk, two-pass epilogue, missing ranks staged from the tile's own edge rank) against k_conv3x3_g16: the same bits -- one
    to seven tiles per workgroup, fewer tiles than workgroups, with and without residual / ReLU, both tile orders, output over the
    residual, the middle launch of the edge-pair form, and the planned boundary cut into parts."""
    from chinesechesszero_amd import _lib
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(700 + boards)
    cl = torch.channels_last
    L = _lib.lib()
    s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    x = torch.relu(torch.randn(boards, 256, 10, 9, generator=g) * 0.7).to(dev).half().contiguous(memory_format=cl)
    r = (torch.randn(boards, 256, 10, 9, generator=g) * 0.7).to(dev).half().contiguous(memory_format=cl)
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.03).to(dev).half().contiguous(memory_format=cl)
    b = (torch.randn(256, generator=g) * 0.2).to(dev)
    xg, rg = _to_g16(x), _to_g16(r)
    wp = _pack_w(w.permute(0, 2, 3, 1), 256)
    P = _lib.CONV_G16 | _lib.CONV_G16_PERSISTENT | (nwg << 16)

    def run(flags, res, y):
        _lib.check(L.ccz_conv3x3_c256_f16(s, C.c_void_p(xg.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(b.data_ptr()),
                                          C.c_void_p(res.data_ptr()) if res is not None else None, C.c_void_p(y.data_ptr()), boards * 90, flags))
        return y

    for res, relu in ((None, 1), (rg, 1), (None, 0), (rg, 3), (None, 3)):
        want = run(relu | _lib.CONV_G16, res, torch.full_like(xg, float("nan")))
        got = run(relu | P, res, torch.full_like(xg, float("nan")))
        assert torch.equal(got, want), (boards, nwg, res is not None, relu, (got.float() - want.float()).abs().max().item())
        got = run(relu | P | _lib.CONV_G16_EDGE_TILES, res, torch.full_like(xg, float("nan")))   # persistent middle launch + edge-pair launch
        assert torch.equal(got, want), (boards, nwg, res is not None, relu, "edge")
    want = run(1 | _lib.CONV_G16, rg, torch.empty_like(xg))
    y = rg.clone()
    assert torch.equal(run(1 | P, y, y), want)       # output written over the residual (how the tower uses it)
    # the planned boundary: live boards cut into parts, every part a persistent launch
    for live, n_parts in ((0, 1), (1, 1), (boards - 3, 2), (boards, 3)):
        n_live = torch.tensor([live], dtype=torch.int32, device=dev)
        yg = torch.full_like(xg, float("nan"))
        cap = -(-(boards // 16) // n_parts) * 1440
        for part in range(n_parts):
            _lib.check(L.ccz_conv3x3_c256_f16_live(s, C.c_void_p(xg.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(rg.data_ptr()),
                                                   C.c_void_p(yg.data_ptr()), cap, 1 | P | (2 if part & 1 else 0), C.c_void_p(n_live.data_ptr()), part, n_parts))
        groups = -(-live // 16)
        assert torch.equal(yg[:groups], want[:groups])
        assert torch.isnan(yg[groups:].float()).all()
    torch.cuda.synchronize()


@pytest.mark.parametrize("edge", [0, 1])
@pytest.mark.parametrize("live,n_parts", [(0, 1), (1, 1), (16, 2), (17, 1), (100, 3), (160, 4)])
def test_group_of_16_kernel_live_rows(live, n_parts, edge):
    """The planned boundary in the group-of-16 layout: the first `live` boards (a device value) are cut into n_parts equal ranges
    of whole groups; the parts together compute exactly the groups that hold live boards, nothing else is written."""
    from chinesechesszero_amd import _lib
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(900 + live)
    cl = torch.channels_last
    L = _lib.lib()
    s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    boards = 160
    x = torch.relu(torch.randn(boards, 256, 10, 9, generator=g) * 0.7).to(dev).half().contiguous(memory_format=cl)
    r = (torch.randn(boards, 256, 10, 9, generator=g) * 0.7).to(dev).half().contiguous(memory_format=cl)
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.03).to(dev).half().contiguous(memory_format=cl)
    b = (torch.randn(256, generator=g) * 0.2).to(dev)
    xg, rg = _to_g16(x), _to_g16(r)
    w = _pack_w(w.permute(0, 2, 3, 1), 256)
    full = torch.empty_like(xg)
    _lib.check(L.ccz_conv3x3_c256_f16(s, C.c_void_p(xg.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(rg.data_ptr()),
                                      C.c_void_p(full.data_ptr()), boards * 90, 1 | _lib.CONV_G16))
    n_live = torch.tensor([live], dtype=torch.int32, device=dev)
    yg = torch.full_like(xg, float("nan"))
    cap = -(-(boards // 16) // n_parts) * 1440
    for part in range(n_parts):
        _lib.check(L.ccz_conv3x3_c256_f16_live(s, C.c_void_p(xg.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(rg.data_ptr()),
                                               C.c_void_p(yg.data_ptr()), cap, 1 | _lib.CONV_G16 | (2 if part & 1 else 0) | (_lib.CONV_G16_EDGE_TILES if edge else 0), C.c_void_p(n_live.data_ptr()), part, n_parts))
    groups = -(-live // 16)
    assert torch.equal(yg[:groups], full[:groups])
    assert torch.isnan(yg[groups:].float()).all()


@pytest.mark.parametrize("edge", [0, 1])
@pytest.mark.parametrize("boards,live,n_parts", [(16, None, 1), (80, None, 1), (96, None, 1), (160, 0, 1), (160, 1, 1), (160, 17, 1),
                                                 (160, 100, 3), (160, 160, 4), (160, 150, 2)])
def test_last_layer_with_heads_in_its_epilogue_is_bit_identical(boards, live, n_parts, edge):
    """ccz_conv3x3_c256_heads_f16 (the tower's last layer, its output never stored, both head convolutions on the rows while they
    are in LDS) writes the same bits as the layer followed by ccz_heads_conv1x1_f16 -- whole batch, an odd number of groups (the
    edge kernel's duplicated half), and the planned boundary cut into parts; boards past the live count are left alone."""
    from chinesechesszero_amd import _lib
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(1700 + boards + (live or 0))
    cl = torch.channels_last
    L = _lib.lib()
    s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    x = torch.relu(torch.randn(boards, 256, 10, 9, generator=g) * 0.7).to(dev).half().contiguous(memory_format=cl)
    r = torch.relu(torch.randn(boards, 256, 10, 9, generator=g) * 0.7).to(dev).half().contiguous(memory_format=cl)
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.03).to(dev).half().contiguous(memory_format=cl)
    b = (torch.randn(256, generator=g) * 0.2).to(dev)
    w32 = torch.zeros(32, 256, dtype=torch.float16)
    w32[:24] = (torch.randn(24, 256, generator=g) * 0.08).half()
    b32 = torch.zeros(32)
    b32[:24] = torch.randn(24, generator=g) * 0.2
    w32, b32 = w32.to(dev), b32.to(dev)
    xg, rg = _to_g16(x), _to_g16(r)
    wp = _pack_w(w.permute(0, 2, 3, 1), 256)
    fl = 1 | _lib.CONV_G16 | (_lib.CONV_G16_EDGE_TILES if edge else 0)
    full = torch.empty_like(xg)
    _lib.check(L.ccz_conv3x3_c256_f16(s, P(xg), P(wp), P(b), P(rg), P(full), boards * 90, fl))
    n_live = None if live is None else torch.tensor([live], dtype=torch.int32, device=dev)
    pol0 = torch.full((boards, 1536), 3.0, dtype=torch.float16, device=dev)
    val0 = torch.full((boards, 640), 3.0, dtype=torch.float16, device=dev)
    _lib.check(L.ccz_heads_conv1x1_f16(s, P(full), P(w32), P(b32), P(pol0), P(val0), boards, _lib.CONV_G16, None if live is None else P(n_live)))
    pol1, val1 = torch.full_like(pol0, 3.0), torch.full_like(val0, 3.0)
    keep = rg.clone()
    if live is None:
        _lib.check(L.ccz_conv3x3_c256_heads_f16(s, P(xg), P(wp), P(b), P(rg), P(w32), P(b32), P(pol1), P(val1), boards * 90, fl, None, 0, 1))
    else:
        cap = -(-(boards // 16) // n_parts) * 1440
        for part in range(n_parts):
            _lib.check(L.ccz_conv3x3_c256_heads_f16(s, P(xg), P(wp), P(b), P(rg), P(w32), P(b32), P(pol1), P(val1), cap, fl | (2 if part & 1 else 0),
                                                    P(n_live), part, n_parts))
    torch.cuda.synchronize()
    assert torch.equal(rg, keep)   # the residual operand is read, nothing is stored over it
    assert torch.equal(pol1, pol0) and torch.equal(val1, val0)
    n = boards if live is None else live
    assert float((pol1[n:] - 3.0).abs().max() if n < boards else 0) == 0 and float((pol1[:n, 1530:] - 3.0).abs().max() if n else 0) == 0
    if n:
        assert float((pol1[:n, :1530] - 3.0).abs().max()) > 0
    # argument checks: board-major rows, no residual
    assert L.ccz_conv3x3_c256_heads_f16(s, P(xg), P(wp), P(b), P(rg), P(w32), P(b32), P(pol1), P(val1), boards * 90, 1, None, 0, 1) != 0
    assert L.ccz_conv3x3_c256_heads_f16(s, P(xg), P(wp), P(b), None, P(w32), P(b32), P(pol1), P(val1), boards * 90, fl, None, 0, 1) != 0


@pytest.mark.parametrize("M,live", [(1040, None), (4096, None), (4096, 3660), (4096, 257), (4000, 1), (300, None), (11, None), (16, None), (16, 5), (1, None)])
def test_both_fc_kernels_give_the_same_bits(M, live):
    """ccz_fc_f16 picks the 256 x 144 tile kernel (k_fc_wide_f16) for many rows x thousands of columns, the one-wave-per-16-columns
    kernel (k_fc_skinny_f16) for up to 16 rows and the 128 x 128 one (k_fc_f16) otherwise; relu bits 1 / 2 force one of the tile kernels. Same operands, same chain of MFMAs per output element: the same bits, with and
    without a device-side live count, ReLU or not, for the policy shape and (forced) the value shape; rows past the live count and
    the columns past n are left alone."""
    from chinesechesszero_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(M + (live or 0))
    s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    n_live = None if live is None else torch.tensor([live], dtype=torch.int32, device=dev)
    for K, N, relu in ((1536, 2086, 0), (640, 256, 1), (128, 1150, 1)):
        Np = -(-N // 128) * 128
        a = torch.relu(torch.randn(M, K, generator=g)).half().to(dev)
        w = torch.zeros(Np, K, dtype=torch.float16)
        w[:N] = (torch.randn(N, K, generator=g) * 0.03).half()
        w = w.to(dev)
        b = torch.zeros(Np)
        b[:N] = torch.randn(N, generator=g) * 0.3
        b = b.to(dev)
        ldc = N + 10
        outs = []
        for force in (2, 4, 0):
            c = torch.full((M, ldc), 9.0, dtype=torch.float16, device=dev)
            _lib.check(L.ccz_fc_f16(s, P(a), K, P(w), P(b), P(c), ldc, M, N, K, relu | force, None if live is None else P(n_live)))
            outs.append(c)
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (K, N)
        n = M if live is None else live
        assert float((outs[1][:, N:] - 9.0).abs().max()) == 0 and (n == M or float((outs[1][n:] - 9.0).abs().max()) == 0)
        ref = a[:n].float() @ w[:N].float().t() + b[:N]
        ref = torch.relu(ref) if relu else ref
        assert torch.allclose(outs[1][:n, :N].float(), ref, atol=1.5e-2, rtol=4e-3), (K, N)


@pytest.mark.parametrize("B,g16", [(1, False), (7, False), (200, False), (48, True), (1040, True)])
def test_head_kernels_against_float32(B, g16):
    """csrc/cczero_heads.h against plain float32 torch: both 1x1 head convolutions + bias + ReLU with the board-order output
    (k_head_conv1x1, either row layout), the FC GEMM with bias / ReLU and K, N tails (k_fc_f16), value_fc2 + tanh (k_value_out);
    a device-side live-row count leaves the rows past it untouched."""
    import ctypes as C
    from chinesechesszero_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(B)
    s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    x = torch.relu(torch.randn(B, 90, 256, generator=g)).half()                       # board-major rows [board][pos][256]
    w = (torch.randn(24, 256, generator=g) * 0.08).half()
    b = (torch.randn(24, generator=g) * 0.2).float()
    w32 = torch.zeros(32, 256, dtype=torch.float16)
    w32[:24] = w
    b32 = torch.zeros(32)
    b32[:24] = b
    want = torch.relu(x.float() @ w.float().t() + b)                                  # [B, 90, 24]
    xr = x.view(B // 16, 16, 90, 256).permute(0, 2, 1, 3).contiguous() if g16 else x  # group-of-16 rows: (g * 90 + pos) * 16 + j
    pol = torch.zeros(B, 1536, dtype=torch.float16, device=dev)
    val = torch.zeros(B, 640, dtype=torch.float16, device=dev)
    xd, wd, bd = xr.to(dev).contiguous(), w32.to(dev), b32.to(dev)      # (named: a temporary would be recycled before the launch runs)
    _lib.check(L.ccz_heads_conv1x1_f16(s, P(xd), P(wd), P(bd), P(pol), P(val), B, _lib.CONV_G16 if g16 else 0, None))
    got_p, got_v = pol.cpu().float(), val.cpu().float()
    assert torch.allclose(got_p[:, :1530].view(B, 90, 17), want[:, :, :17], atol=2e-3, rtol=2e-3)
    assert torch.allclose(got_v[:, :630].view(B, 90, 7), want[:, :, 17:], atol=2e-3, rtol=2e-3)
    assert float(got_p[:, 1530:].abs().max()) == 0 and float(got_v[:, 630:].abs().max()) == 0      # pad columns are never written
    # with a live count: boards past it keep what they held
    live_n = max(1, B // 2)
    live = torch.tensor([live_n], dtype=torch.int32, device=dev)
    pol2 = torch.full((B, 1536), 7.0, dtype=torch.float16, device=dev)
    val2 = torch.full((B, 640), 7.0, dtype=torch.float16, device=dev)
    _lib.check(L.ccz_heads_conv1x1_f16(s, P(xd), P(wd), P(bd), P(pol2), P(val2), B, _lib.CONV_G16 if g16 else 0, P(live)))
    assert torch.equal(pol2[:live_n, :1530], pol[:live_n, :1530]) and float((pol2[live_n:] - 7.0).abs().max() if live_n < B else 0) == 0
    # ---- the FC GEMM: policy shape (K 1536, N 2086 = 16 tiles + a 38-column tail) and value shape (K 640, N 256, ReLU)
    for K, N, relu in ((1536, 2086, 0), (640, 256, 1)):
        Np = -(-N // 128) * 128
        a = torch.zeros(B, K, dtype=torch.float16)
        a[:, :K - 6] = torch.relu(torch.randn(B, K - 6, generator=g)).half()
        wf = torch.zeros(Np, K, dtype=torch.float16)
        wf[:N, :K - 6] = (torch.randn(N, K - 6, generator=g) * 0.03).half()
        bf = torch.zeros(Np)
        bf[:N] = torch.randn(N, generator=g) * 0.3
        ref = a.float() @ wf[:N].float().t() + bf[:N]
        ref = torch.relu(ref) if relu else ref
        c = torch.full((B, N), 9.0, dtype=torch.float16, device=dev)
        ad, wfd, bfd = a.to(dev), wf.to(dev), bf.to(dev)
        _lib.check(L.ccz_fc_f16(s, P(ad), K, P(wfd), P(bfd), P(c), N, B, N, K, relu, None))
        assert torch.allclose(c.cpu().float(), ref, atol=1.5e-2, rtol=4e-3), (K, N)
        c2 = torch.full((B, N), 9.0, dtype=torch.float16, device=dev)
        _lib.check(L.ccz_fc_f16(s, P(ad), K, P(wfd), P(bfd), P(c2), N, B, N, K, relu, P(live)))
        assert torch.equal(c2[:live_n], c[:live_n]) and (live_n == B or float((c2[live_n:] - 9.0).abs().max()) == 0)
    # ---- value_fc2 + tanh
    h = torch.relu(torch.randn(B, 256, generator=g)).half()
    w2 = (torch.randn(256, generator=g) * 0.1).half()
    v = torch.full((B,), 5.0, device=dev)
    hd, w2d = h.to(dev), w2.to(dev)
    _lib.check(L.ccz_value_out_f32(s, P(hd), P(w2d), 0.125, P(v), B, None))
    refv = torch.tanh((h.float() @ w2.float() + 0.125).half().float())
    assert torch.allclose(v.cpu(), refv, atol=2e-3)
