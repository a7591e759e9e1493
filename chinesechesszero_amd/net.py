"""Policy-value network and the evaluator boundary (mirror of reference net.py).

The network stays in PyTorch-ROCm (north star: "the net stays in PyTorch"). ``Net`` re-declares the
reference architecture with identical ``state_dict`` keys (net.py:46-110) so reference-trained
``.pkl`` weights load. What changes is the boundary: instead of one batch-1 call per playout
(net.py:151-205) the engine hands over ONE contiguous fp16 batch of all leaves per lockstep step
(:meth:`PolicyValueNet.evaluate_leaves`), on the same HIP stream as the simulator kernels.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

PIECES = 7   # net.py:11
PLAYS = 17   # net.py:12  red 8 + black 8 + side-to-move


class ResBlock(nn.Module):
    """net.py:15-43"""

    def __init__(self, num_channels=256):
        super().__init__()
        self.conv1 = nn.Conv2d(num_channels, num_channels, kernel_size=(3, 3), stride=(1, 1), padding=1)
        self.conv1_bn = nn.BatchNorm2d(num_channels)
        self.conv1_act = nn.ReLU()
        self.conv2 = nn.Conv2d(num_channels, num_channels, kernel_size=(3, 3), stride=(1, 1), padding=1)
        self.conv2_bn = nn.BatchNorm2d(num_channels)
        self.conv2_act = nn.ReLU()

    def forward(self, x):
        y = self.conv1_act(self.conv1_bn(self.conv1(x)))
        y = self.conv2_bn(self.conv2(y))
        return self.conv2_act(x + y)


class Net(nn.Module):
    """net.py:46-110: input [N,17,7,10,9] -> (log_softmax policy [N,2086], tanh value [N,1])."""

    def __init__(self, num_channels=256, resblocks_num=40):
        super().__init__()
        self.input_channels = PLAYS * PIECES
        self.conv_block = nn.Conv2d(self.input_channels, num_channels, kernel_size=(3, 3), stride=(1, 1), padding=1)
        self.conv_block_bn = nn.BatchNorm2d(num_channels)
        self.conv_block_act = nn.ReLU()
        self.res_blocks = nn.ModuleList([ResBlock(num_channels=num_channels) for _ in range(resblocks_num)])
        self.policy_conv = nn.Conv2d(num_channels, PLAYS, kernel_size=(1, 1), stride=(1, 1))
        self.policy_bn = nn.BatchNorm2d(PLAYS)
        self.policy_act = nn.ReLU()
        self.policy_fc = nn.Linear(PLAYS * 10 * 9, 2086)
        self.value_conv = nn.Conv2d(num_channels, PIECES, kernel_size=(1, 1), stride=(1, 1))
        self.value_bn = nn.BatchNorm2d(PIECES)
        self.value_act1 = nn.ReLU()
        self.value_fc1 = nn.Linear(PIECES * 10 * 9, 256)
        self.value_act2 = nn.ReLU()
        self.value_fc2 = nn.Linear(256, 1)

    def forward(self, x):
        x = x.view(x.shape[0], -1, 10, 9)
        x = self.conv_block_act(self.conv_block_bn(self.conv_block(x)))
        for blk in self.res_blocks:
            x = blk(x)
        policy = self.policy_act(self.policy_bn(self.policy_conv(x)))
        policy = F.log_softmax(self.policy_fc(torch.reshape(policy, [-1, PLAYS * 10 * 9])), dim=1)
        value = self.value_act1(self.value_bn(self.value_conv(x)))
        value = self.value_act2(self.value_fc1(torch.reshape(value, [-1, PIECES * 10 * 9])))
        value = torch.tanh(self.value_fc2(value))
        return policy, value


def _fold(conv: nn.Conv2d, bn: nn.BatchNorm2d):
    """eval-mode BatchNorm folded into the preceding conv (exact in real arithmetic)."""
    scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    w = conv.weight.detach().double() * scale.view(-1, 1, 1, 1)
    b = (conv.bias.detach().double() - bn.running_mean.detach().double()) * scale + bn.bias.detach().double()
    return w, b


def pack_conv_weights_g16(w):
    """[256, 3, 3, cin] (channels-last memory order of a [co, ci, 3, 3] weight) -> [cin / 32, 9, 256, 32] with the four 16-byte
    chunks of every 64-byte row in the swizzled order of the kernel's LDS image (position pos of row co holds chunk
    pos ^ ((-(co >> 2)) & 3)): what ``k_conv3x3_g16`` reads as contiguous 16 KB half-tiles (csrc/cczero_conv_g16.h)."""
    co, cin = w.shape[0], w.shape[-1]
    w5 = w.reshape(co, 9, cin // 32, 4, 8)
    f = (-(torch.arange(co, device=w.device) >> 2)) & 3
    idx = torch.arange(4, device=w.device)[None, :] ^ f[:, None]
    w5 = torch.gather(w5, 3, idx[:, None, None, :, None].expand(co, 9, cin // 32, 4, 8))
    return w5.permute(2, 1, 0, 3, 4).contiguous().view(cin // 32, 9, co, 32)


class EvalOptions:
    """The evaluator's A/B switches, read from the environment ONCE (when an ``InferenceNet`` is built) instead of on every
    call of the hot host path; non-default values are logged once. Tests and A/B scripts change them on a live object with
    ``InferenceNet.set_options(layout="g16", ...)``.

    ``CCZ_FUSED_CONV=0`` (fused_conv): MIOpen convolutions + one-pass epilogue instead of the hand-written tower kernels;
    ``CCZ_FUSED_STEM=0`` (fused_stem): the stem through torch; ``CCZ_FUSED_HEADS=0`` (fused_heads): heads and FC layers through
    torch GEMMs; ``CCZ_CONV_LAYOUT=auto|nhwc|g16`` (layout): activation row layout (auto: group-of-16 from 640 boards on);
    ``CCZ_CONV_FORCE=small|tile`` (force): one convolution kernel whatever the batch size; ``CCZ_TOWER_GROUPS`` / ``CCZ_TOWER_CHAINS``
    (groups, chains): launch structure of the tower; ``CCZ_CONV_ZIGZAG=0`` (zigzag): no alternating tile order;
    ``CCZ_CONV_EDGE_TILES=0|1|auto`` (edge_tiles): the group-of-16 convolution as a middle launch + an edge-pair launch (six live taps on
    the edge ranks; same values). auto (default): from 4096 boards on, with THREE launch chains -- +0.7...0.8 % sims/s there (three
    interleaved pairs on one box, profiles/r04_conv_g16.json); below, the launches get too small: -1.4 % at 3072 boards, -5 % at 2048,
    -14 % at 1024 (where round 3's single launch per layer stays);
    ``CCZ_FUSED_LAST=0`` (fused_last): the head convolutions as a pass of their own over the stored output of the tower instead of in
    the last layer's epilogue (group-of-16 rows; same bits either way);
    ``CCZ_CONV_PERSISTENT=N`` (persistent; round 6, A/B): the group-of-16 tower layers on N persistent workgroups per launch that walk
    tile lists (k_conv3x3_g16_pers, csrc/cczero_conv_g16p.h; same bits); 0 = one tile per workgroup (default)."""

    FIELDS = ("fused_conv", "fused_stem", "fused_heads", "fused_last", "layout", "force", "groups", "chains", "zigzag", "edge_tiles", "persistent")

    def __init__(self, env=None):
        env = os.environ if env is None else env
        self.fused_conv = env.get("CCZ_FUSED_CONV", "1") != "0"
        self.fused_stem = env.get("CCZ_FUSED_STEM", "1") != "0"
        self.fused_heads = env.get("CCZ_FUSED_HEADS", "1") != "0"
        self.fused_last = env.get("CCZ_FUSED_LAST", "1") != "0"
        self.layout = env.get("CCZ_CONV_LAYOUT", "auto")
        self.force = {"small": 16, "tile": 32}.get(env.get("CCZ_CONV_FORCE", ""), 0)
        self.groups = int(env.get("CCZ_TOWER_GROUPS", "0"))
        self.chains = int(env.get("CCZ_TOWER_CHAINS", "0"))   # 0 = InferenceNet.TOWER_CHAINS
        self.zigzag = env.get("CCZ_CONV_ZIGZAG", "1") == "1"
        # group-of-16 layout: ranks 0 / 9 on the edge-pair kernel (round 4): "auto" = from 4096 boards on, together with three launch chains
        self.edge_tiles = {"0": False, "1": True}.get(env.get("CCZ_CONV_EDGE_TILES", "auto"), "auto")
        self.persistent = int(env.get("CCZ_CONV_PERSISTENT", "0"))
        if self.layout not in ("auto", "nhwc", "g16"):
            raise ValueError("CCZ_CONV_LAYOUT must be auto, nhwc or g16")

    def non_default(self):
        d = EvalOptions(env={})
        return {f: getattr(self, f) for f in self.FIELDS if getattr(self, f) != getattr(d, f)}


_CHAIN_STREAMS: dict = {}


def chain_streams(device):
    """The tower's launch-chain streams of ``device``, one set per process: created on the first call with a first (tiny) piece of work
    on each, so that the HIP runtime binds them to hardware queues THEN. Which queue a stream gets depends on the ORDER of first use
    across the whole process (HIP: GPU_MAX_HW_QUEUES queues, later streams share); measured on one MI355X
    (profiles/r05_hwq_probe.txt): chains bound after the multi-GPU exchange's streams were in use cost the evaluator +6 % (eight or
    sixteen queues) to +21 % (HIP's default four), chains bound before them nothing. ``PolicyValueNet.refresh_inference_copy`` calls
    this as soon as the first inference copy is on the device -- in every flow before the exchange is first used."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    got = _CHAIN_STREAMS.get(device)
    if got is None:
        got = [torch.cuda.Stream(device=device) for _ in range(7)]
        cur = torch.cuda.current_stream(device)
        ev = torch.cuda.Event()
        ev.record(cur)
        for st in got:
            st.wait_event(ev)
            with torch.cuda.stream(st):
                torch.zeros(1, device=device).add_(1.0)       # a real dispatch: the queue exists from here on
            done = torch.cuda.Event()
            done.record(st)
            cur.wait_event(done)
        _CHAIN_STREAMS[device] = got
    return got


class InferenceNet(nn.Module):
    """Inference copy of ``Net`` for the lockstep evaluator: BN folded, fp16, channels-last.

    The first conv only sees the 21 planes that can be non-zero on the search path (groups 7, 15, 16;
    net.py:160-173 leaves the other 14 groups zero), which is exact: zero inputs contribute nothing.
    Outputs: prob = exp(log_softmax(policy)) float32 [B,2086] (net.py:202), value float32 [B].
    """

    LIVE = list(range(7 * 7, 8 * 7)) + list(range(15 * 7, 17 * 7))  # channel ids of groups 7, 15, 16

    def __init__(self, net: Net, dtype=torch.float16, live_only: bool = True):
        super().__init__()
        self.dtype = dtype
        self.live_only = live_only
        self.opt = EvalOptions()
        if self.opt.non_default():
            from .tools import log
            log(f"evaluator options from the environment: {self.opt.non_default()}")
        cl = torch.channels_last
        w, b = _fold(net.conv_block, net.conv_block_bn)
        if live_only:
            w = w[:, self.LIVE]
        self.stem_w = nn.Parameter(w.to(dtype).contiguous(memory_format=cl), requires_grad=False)
        self.stem_b = nn.Parameter(b.to(dtype), requires_grad=False)
        if live_only and w.shape[0] == 256:
            # stem weights for the fused kernel: [co, ky, kx, 64] with the 21 live input channels first, zeros above
            w64 = torch.zeros(256, 3, 3, 64, dtype=w.dtype, device=w.device)
            w64[..., :len(self.LIVE)] = w.permute(0, 2, 3, 1)
            self.stem_w64 = nn.Parameter(w64.to(dtype).contiguous(), requires_grad=False)
            self.stem_b32 = nn.Parameter(b.detach().float().clone(), requires_grad=False)
        ws, bs = [], []
        for blk in net.res_blocks:
            for conv, bn in ((blk.conv1, blk.conv1_bn), (blk.conv2, blk.conv2_bn)):
                w, b = _fold(conv, bn)
                ws.append(nn.Parameter(w.to(dtype).contiguous(memory_format=cl), requires_grad=False))
                bs.append(nn.Parameter(b.to(dtype), requires_grad=False))
        self.ws = nn.ParameterList(ws)
        self.bs = nn.ParameterList(bs)
        # the same weights packed for k_conv3x3_g16 (contiguous half-tiles; ccz_pack_conv_weights_g16_f16 is the C-side twin)
        if ws and ws[0].shape[0] == 256 and ws[0].shape[1] == 256:
            self.ws_g16 = nn.ParameterList([nn.Parameter(pack_conv_weights_g16(w.permute(0, 2, 3, 1)), requires_grad=False) for w in ws])
        if hasattr(self, "stem_w64"):
            self.stem_w64_g16 = nn.Parameter(pack_conv_weights_g16(self.stem_w64), requires_grad=False)
        # float32 copies of the tower biases for the fused convolution kernel (bias is added to the fp32 accumulator)
        self.bs32 = nn.ParameterList([nn.Parameter(b.detach().float().clone(), requires_grad=False) for b in bs])
        # both 1x1 head convs as ONE plain GEMM on the NHWC rows: [pixels, C] x [C, 17 policy + 7 value channels]. (As an
        # MIOpen 1x1 convolution the same product was NOT bit-reproducible from call to call below ~300 boards --
        # profiles/determinism_probe.py -- which made small-batch searches irreproducible; a GEMM is, and it leaves its
        # output pixel-major, so the FC weights are stored with their input columns permuted from the reference's
        # (channel, pixel) flatten order (net.py:98,103) to (pixel, channel) instead of copying activations around.)
        wp, bp = _fold(net.policy_conv, net.policy_bn)
        wv, bv = _fold(net.value_conv, net.value_bn)
        C_in = wp.shape[1]
        self.head_wT = nn.Parameter(torch.cat([wp, wv], 0).reshape(PLAYS + PIECES, C_in).t().contiguous().to(dtype), requires_grad=False)
        self.head_b = nn.Parameter(torch.cat([bp, bv], 0).to(dtype), requires_grad=False)
        pc = lambda w, c: w.detach().reshape(w.shape[0], c, 90).permute(0, 2, 1).reshape(w.shape[0], c * 90).contiguous()
        self.policy_fc_w = nn.Parameter(pc(net.policy_fc.weight, PLAYS).to(dtype), requires_grad=False)
        self.policy_fc_b = nn.Parameter(net.policy_fc.bias.detach().to(dtype), requires_grad=False)
        self.value_fc1_w = nn.Parameter(pc(net.value_fc1.weight, PIECES).to(dtype), requires_grad=False)
        self.value_fc1_b = nn.Parameter(net.value_fc1.bias.detach().to(dtype), requires_grad=False)
        self.value_fc2_w = nn.Parameter(net.value_fc2.weight.detach().to(dtype), requires_grad=False)
        self.value_fc2_b = nn.Parameter(net.value_fc2.bias.detach().to(dtype), requires_grad=False)
        # the same weights laid out for the hand-written head kernels (csrc/cczero_heads.h; include/cczero.h ccz_heads_conv1x1_f16,
        # ccz_fc_f16, ccz_value_out_f32): head convolutions as ONE [32, 256] matrix (17 policy + 7 value rows + 8 zero rows), FC
        # weights with their input columns in (pos, channel) order zero-padded to the kernels' K granule (1530 -> 1536, 630 -> 640)
        # and their rows to whole 128-row tiles, biases in float32 (added to the fp32 accumulator)
        if C_in == 256 and dtype == torch.float16:
            w32 = torch.zeros(32, C_in, dtype=torch.float64)
            w32[:PLAYS + PIECES] = torch.cat([wp, wv], 0).reshape(PLAYS + PIECES, C_in)
            b32 = torch.zeros(32, dtype=torch.float64)
            b32[:PLAYS + PIECES] = torch.cat([bp, bv], 0)
            self.head_w32 = nn.Parameter(w32.to(dtype), requires_grad=False)
            self.head_b32 = nn.Parameter(b32.float(), requires_grad=False)

            def padded(w, rows, cols):
                out = torch.zeros(rows, cols, dtype=dtype)
                out[:w.shape[0], :w.shape[1]] = w.to(dtype)
                return out

            def padded_bias(b, rows):
                out = torch.zeros(rows, dtype=torch.float32)
                out[:b.shape[0]] = b.detach().to(dtype).float()   # (the fp16-rounded bias the torch path adds)
                return out
            self.policy_fc_wp = nn.Parameter(padded(pc(net.policy_fc.weight, PLAYS), 2176, 1536), requires_grad=False)
            self.policy_fc_b32 = nn.Parameter(padded_bias(net.policy_fc.bias, 2176), requires_grad=False)
            self.value_fc1_wp = nn.Parameter(padded(pc(net.value_fc1.weight, PIECES), 256, 640), requires_grad=False)
            self.value_fc1_b32 = nn.Parameter(padded_bias(net.value_fc1.bias, 256), requires_grad=False)

    def set_options(self, **kw):
        """Change A/B switches of a live object (tests, profile scripts): ``set_options(layout="g16", fused_conv=False)``;
        ``force`` takes "small" / "tile" / "" as the environment variable does."""
        for k, v in kw.items():
            if k not in EvalOptions.FIELDS:
                raise KeyError(k)
            if k == "force" and isinstance(v, str):
                v = {"small": 16, "tile": 32, "": 0}[v]
            setattr(self.opt, k, v)
        if self.opt.layout not in ("auto", "nhwc", "g16"):
            raise ValueError("layout must be auto, nhwc or g16")
        self.__dict__.pop("_path_cache", None)
        return self

    def opt_fused_conv(self) -> bool:
        return self.opt.fused_conv

    def tower_groups(self, B: int, g16: bool) -> int:
        """Sequential board groups of the tower for a batch of B boards (Infinity-Cache residency, :meth:`_tower_fused`)."""
        return self.opt.groups or -(-B // (self.TOWER_GROUP_BOARDS_G16 if g16 else self.TOWER_GROUP_BOARDS))

    EDGE_TILES_MIN_BOARDS = 4096   # "auto": the edge-pair kernel + three chains from here on (measured crossover between 3072 and 4096)
    TOWER_CHAINS_EDGE = 3

    def _edge(self, B: int, g16: bool = True) -> bool:
        """Does a batch of B boards run the group-of-16 convolution as middle launch + edge-pair launch (cczero_conv_g16e.h)?"""
        et = self.opt.edge_tiles
        return bool(g16 and (et is True or (et == "auto" and B >= self.EDGE_TILES_MIN_BOARDS)))

    TOWER_CHAINS_SMALL = 3          # ... and three for batches whose chains are a single under-filled round of tiles each
    TOWER_CHAINS_SMALL_BOARDS = (768, 1536)

    def tower_chains(self, B: int, groups: int = 1, edge: bool = False) -> int:
        """Concurrent launch chains per group (one HIP stream each). Two by default, three with the edge-pair kernel (4096 boards
        on) -- and three around 1024 boards (BASELINE configs[1]): ~290 live tiles are 1.13 rounds of the 256 CUs, every chain's layer
        is one partial round and pays the dependent-launch gap in full, and three chains keep the chip fuller than two: 178.2 k
        against 176.0 k sims/s, 4 chains 177.5 k, 1 chain 126.6 k (profiles/r06_chains_1024.txt, one box, two interleaved
        repetitions; 2048 boards: 2 chains 191.8 k, 3: 189.0 k, 4: 183.8 k -- two stay)."""
        per_group = -(-B // groups)
        default = self.TOWER_CHAINS_EDGE if edge else (self.TOWER_CHAINS_SMALL if self.TOWER_CHAINS_SMALL_BOARDS[0] <= per_group < self.TOWER_CHAINS_SMALL_BOARDS[1]
                                                       else self.TOWER_CHAINS)
        return max(1, min(self.opt.chains or default, 8, per_group // 256))

    def bind_chain_streams(self, device):
        """Give this inference copy the process-wide launch-chain streams of ``device`` (:func:`chain_streams`): created and first used
        ONCE per process, as early as the first inference copy is built, and reused by every later copy (a weight reload builds a new
        ``InferenceNet``; new streams at that point would be bound AFTER the exchange's)."""
        device = torch.device(device)
        if device.type == "cuda":
            self._chain_streams = (device, chain_streams(device))

    def derived_parameter_names(self) -> set:
        """Names (as in ``named_parameters()``) of the parameters that are pure functions of OTHER parameters of this module -- the
        weights packed for k_conv3x3_g16. ``replay.broadcast_model(what="inference")`` does not send them and :meth:`repack_derived`
        rebuilds exactly this set: one list for both, so a future packed tensor cannot be skipped by one and forgotten by the other."""
        names = set()
        if hasattr(self, "ws_g16"):
            names |= {f"ws_g16.{i}" for i in range(len(self.ws_g16))}
        if hasattr(self, "stem_w64_g16"):
            names.add("stem_w64_g16")
        return names

    @torch.no_grad()
    def repack_derived(self):
        """The tensors derived from others (the weights packed for k_conv3x3_g16) re-derived IN PLACE after the primary ones were
        overwritten (``replay.broadcast_model(what="inference")``): addresses stay what captured graphs hold."""
        done = set()
        if hasattr(self, "ws_g16"):
            for i, (dst, w) in enumerate(zip(self.ws_g16, self.ws)):
                dst.copy_(pack_conv_weights_g16(w.permute(0, 2, 3, 1)))
                done.add(f"ws_g16.{i}")
        if hasattr(self, "stem_w64_g16"):
            self.stem_w64_g16.copy_(pack_conv_weights_g16(self.stem_w64))
            done.add("stem_w64_g16")
        if done != self.derived_parameter_names():
            raise RuntimeError(f"repack_derived rebuilt {sorted(done)} but derived_parameter_names() lists {sorted(self.derived_parameter_names())}")
        self.__dict__.pop("_b2", None)

    def _epilogue(self, y, bias, residual=None):
        """relu(y + bias [+ residual]) in ONE pass (libcczero ccz_bias_act_f16) on NHWC fp16 device tensors;
        plain torch ops otherwise (CPU tests, other dtypes)."""
        if y.is_cuda and y.dtype == torch.float16 and y.is_contiguous(memory_format=torch.channels_last) \
                and (residual is None or residual.is_contiguous(memory_format=torch.channels_last)):
            import ctypes as C
            from . import _lib
            n, c, h, w = y.shape
            _lib.check(_lib.lib().ccz_bias_act_f16(C.c_void_p(torch.cuda.current_stream(y.device).cuda_stream),
                                                   C.c_void_p(y.data_ptr()), C.c_void_p(bias.data_ptr()),
                                                   C.c_void_p(residual.data_ptr()) if residual is not None else None,
                                                   n * h * w, c))
            return y
        y = y + bias.view(1, -1, 1, 1)
        if residual is not None:
            y = y + residual
        return F.relu_(y)

    FUSED_MIN_BOARDS = 1  # every batch runs on the hand-written convolution: up to 64 boards on k_conv3x3_small (16-channel x
    # 64-pixel blocks spread over the chip), above that on k_conv3x3_c256 (256-pixel tiles) and, from 640 boards, k_conv3x3_g16 (group-of-16 rows) -- the same values, so a
    # board's tower activations do not depend on the batch size. (Round 2 sent batches under 192 boards to MIOpen + an epilogue
    # pass: 12.1 us per tower layer at one board, profiles/r03_single_board.json.)

    def _use_fused_tower(self, x) -> bool:
        """The hand-written MFMA convolution (libcczero ccz_conv3x3_c256_f16) covers the tower's shape only:
        fp16 NHWC on the GPU, 256 channels. ``fused_conv=False`` (``CCZ_FUSED_CONV=0``) selects the MIOpen + epilogue path for A/B runs."""
        if not self.opt.fused_conv:
            return False
        return bool(x.is_cuda and x.dtype == torch.float16 and x.shape[1] == 256 and x.shape[0] >= self.FUSED_MIN_BOARDS
                    and x.is_contiguous(memory_format=torch.channels_last))

    G16_MIN_BOARDS = 640  # batches from here on run in the group-of-16 row layout (cczero_conv_g16.h: whole-rank tiles, off-board taps
    # skipped), padded to a multiple of 16 boards; below it (a single under-filled round of tiles: latency, not throughput) the
    # 256-pixel tile kernel's smaller tiles are 1-2 % quicker (128 / 256 / 512 boards: 3.74 / 3.78 / 3.94 against 3.81 / 3.82 /
    # 3.97 ms per step; from 640 boards on the group-of-16 kernel wins: 640 / 768 / 1024 boards 4.15 / 4.58 / 6.0 against 4.34 / 4.94 / 6.4). ``CCZ_CONV_LAYOUT=nhwc`` keeps the board-major rows at every size,
    # ``CCZ_CONV_LAYOUT=g16`` uses the group-of-16 layout from 65 boards on (A/B runs, tests). Same values either way (all three
    # kernels add in the same order).

    def _g16(self, B) -> bool:
        mode = self.opt.layout
        if mode == "nhwc" or self.opt.force:
            return False
        return B >= (65 if mode == "g16" else self.G16_MIN_BOARDS)

    TOWER_GROUP_BOARDS_G16 = 4096
    TOWER_GROUP_BOARDS = 2048  # boards per sequential group (working set of a group fits the Infinity Cache); env CCZ_TOWER_GROUPS
    TOWER_CHAINS = 2  # independent board ranges run as concurrent launch chains (one HIP stream each); env CCZ_TOWER_CHAINS (<= 8); three with the edge-pair kernel (TOWER_CHAINS_EDGE)

    def _tower_fused(self, x, plan=None, g16=None, heads=None):
        """40 residual blocks = 80 launches of one kernel: conv3x3 + bias [+ x] + ReLU each (reference net.py:20-43).
        ``plan`` = (rows, n_rows) of the planned evaluator boundary: only the first ``n_rows`` (a device value) boards of ``x``
        are live; every launch skips the tiles past them (``ccz_conv3x3_c256_f16_live``).

        Boards are independent, so the batch is cut into ``TOWER_CHAINS`` contiguous board ranges whose 80-launch
        chains run on separate HIP streams: the tile tail of one chain's layer is filled by the other chain's tiles. In the bench
        (4096 boards, 80 different weight sets; round 2, board-major rows): 1 chain 29.2 ms/step, 2 chains 27.9, 3: 28.0, 4: 28.5,
        8: 28.6 (same-weights microbench profiles/conv_streams.py: 352 -> 319 -> 311 us per layer for 1 / 2 / 8 chains); round 3,
        group-of-16 rows with the evaluation cache: 1 chain 24.9, 2 chains 23.5, 3: 23.3-24.0, 4: 24.4 (profiles/r03_conv_g16.json).
        Inside a stream capture (hipGraph) one chain is used. ``g16``: the rows of ``x`` are in the group-of-16 order
        (``None``: as :meth:`_stem_fused` lays out a batch of this size). ``heads`` = (pol, val) buffers of :meth:`_head_buffers`
        (group-of-16 rows only): the LAST layer runs as ``ccz_conv3x3_c256_heads_f16`` -- both head convolutions in its epilogue,
        its own output never stored: the returned tensor then holds the input of the last block, not the tower's output."""
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        if g16 is None:   # x as _stem_fused returned it for a batch of this size (padded to whole groups of 16 boards)
            g16 = self._g16(x.shape[0]) and x.shape[0] % 16 == 0
        y = torch.empty_like(x)
        # groups of TOWER_GROUP_BOARDS boards go through all 80 layers one after the other: the two activation buffers of a
        # group (2 x 94 MB at 2048 boards) then stay inside the 256 MB Infinity Cache from layer to layer
        # (4096 boards: 1 group 27.55 ms/step, 2 groups 27.19, 3: 28.0, 4: 28.4)
        Bt = x.shape[0]
        # (group-of-16 layout, round 3, 4096 boards with the evaluation cache: 1 group x 2 chains 23.54 ms/step, 2 x 2 23.80,
        # 1 x 1 24.89, 1 x 4 24.42: the cache-residency gain of two groups is gone, the tail-filling of two chains is not)
        groups = self.tower_groups(Bt, g16)
        if torch.cuda.is_current_stream_capturing():
            groups = 1
        if plan is not None:
            self._tower_planned(L, C, x, y, plan, groups, g16, heads)
            return x
        gstep = -(-(-(-Bt // groups)) // 128) * 128 if groups > 1 else Bt
        for g0 in range(0, Bt, gstep):
            self._tower_range(L, C, x, y, g0, min(Bt, g0 + gstep), g16, heads)
        return x

    def _tower_planned(self, L, C, x, y, plan, groups, g16=False, heads=None):
        """The tower on the LIVE rows of a planned batch (``ccz_eval_plan``): the live rows -- a device-side count -- are cut
        into groups x chains EQUAL ranges by the kernel itself (``ccz_conv3x3_c256_f16_live``: part / n_parts), so that the
        concurrent chains of a group stay balanced whatever the live count is. Launch structure as in :meth:`_tower_range`:
        groups one after the other (Infinity-Cache residency), the chains of a group on separate streams."""
        from . import _lib
        B = x.shape[0]
        cur = torch.cuda.current_stream(x.device)
        edge = self._edge(B, g16)
        chains = 1 if torch.cuda.is_current_stream_capturing() else self.tower_chains(B, groups, edge)
        n_parts = groups * chains
        if g16:
            cap = -(-(B // 16) // n_parts) * 1440            # whole 16-board groups (B is padded to a multiple of 16)
        else:
            cap = -(-(-(-B // n_parts)) // 8) * 8 * 90       # pixels of the largest range a launch may get
        lay = (_lib.CONV_G16 | (_lib.CONV_G16_EDGE_TILES if edge else 0) | self._persistent_flag()) if g16 else 0
        if chains > 1:
            pool = getattr(self, "_chain_streams", None)
            if pool is None or pool[0] != x.device or len(pool[1]) < chains - 1:
                pool = (x.device, chain_streams(x.device))
                self._chain_streams = pool
        live = C.c_void_p(plan[1].data_ptr())
        xp, yp = C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr())
        down = 2 if self.opt.zigzag else 0
        for g in range(groups):
            streams = [cur] + [self._chain_streams[1][k] for k in range(chains - 1)]
            if chains > 1:
                fork = torch.cuda.Event()
                fork.record(cur)
                for st in streams[1:]:
                    st.wait_event(fork)
            wsrc = self.ws_g16 if g16 else self.ws
            for i in range(0, len(self.ws), 2):
                w1, b1_, w2, b2_ = (C.c_void_p(t.data_ptr()) for t in (wsrc[i], self.bs32[i], wsrc[i + 1], self.bs32[i + 1]))
                for k, st in enumerate(streams):
                    _lib.check(L.ccz_conv3x3_c256_f16_live(C.c_void_p(st.cuda_stream), xp, w1, b1_, None, yp, cap, 1 | down | lay, live, g * chains + k, n_parts))
                if heads is not None and i == len(self.ws) - 2:   # the last layer: heads in the epilogue, no output tensor
                    hw, hb, hp, hv = (C.c_void_p(t.data_ptr()) for t in (self.head_w32, self.head_b32, heads[0], heads[1]))
                    for k, st in enumerate(streams):
                        _lib.check(L.ccz_conv3x3_c256_heads_f16(C.c_void_p(st.cuda_stream), yp, w2, b2_, xp, hw, hb, hp, hv, cap, 1 | lay, live, g * chains + k, n_parts))
                    continue
                for k, st in enumerate(streams):
                    _lib.check(L.ccz_conv3x3_c256_f16_live(C.c_void_p(st.cuda_stream), yp, w2, b2_, xp, xp, cap, 1 | lay, live, g * chains + k, n_parts))
            for st in streams[1:]:
                join = torch.cuda.Event()
                join.record(st)
                cur.wait_event(join)

    def _tower_range(self, L, C, x, y, lo, hi, g16=False, heads=None):
        """Boards [lo, hi) through all 80 layers, as TOWER_CHAINS concurrent launch chains."""
        from . import _lib
        B = hi - lo
        cur = torch.cuda.current_stream(x.device)
        edge = self._edge(x.shape[0], g16)
        parts = 1 if torch.cuda.is_current_stream_capturing() else self.tower_chains(B, 1, edge)
        step = -(-B // parts)
        if parts > 1:
            step = -(-step // 128) * 128  # 128 boards = 45 whole tiles: no partial tile inside the batch
        bounds = [(lo + b0, lo + min(B, b0 + step)) for b0 in range(0, B, step)]
        if len(bounds) > 1:
            pool = getattr(self, "_chain_streams", None)
            if pool is None or pool[0] != x.device or len(pool[1]) < len(bounds) - 1:
                pool = (x.device, chain_streams(x.device))
                self._chain_streams = pool
            fork = torch.cuda.Event()
            fork.record(cur)
        row = 90 * 256 * x.element_size()
        chains = []
        for k, (b0, b1) in enumerate(bounds):
            st = cur if k == 0 else self._chain_streams[1][k - 1]
            if k:
                st.wait_event(fork)
            chains.append((st, C.c_void_p(st.cuda_stream), C.c_void_p(x.data_ptr() + b0 * row), C.c_void_p(y.data_ptr() + b0 * row), (b1 - b0) * 90, b0))
        # launches are enqueued layer by layer across the chains, so that the chains advance together (the same layer's
        # weights stay hot in L2) and no chain waits for the host to finish enqueuing another one. The tile order
        # alternates from layer to layer: what the previous layer wrote last (still in the Infinity Cache) is read
        # first (-0.7 % on the step; zigzag=False / CCZ_CONV_ZIGZAG=0 switches it off).
        down = 2 if self.opt.zigzag else 0
        v2 = self.opt.force | ((_lib.CONV_G16 | (_lib.CONV_G16_EDGE_TILES if edge else 0) | self._persistent_flag()) if g16 else 0)
        wsrc = self.ws_g16 if g16 else self.ws
        for i in range(0, len(self.ws), 2):
            w1, b1_, w2, b2_ = (C.c_void_p(t.data_ptr()) for t in (wsrc[i], self.bs32[i], wsrc[i + 1], self.bs32[i + 1]))
            for _, s, xp, yp, n_pixels, _b0 in chains:
                _lib.check(L.ccz_conv3x3_c256_f16(s, xp, w1, b1_, None, yp, n_pixels, 1 | down | v2))
            if heads is not None and i == len(self.ws) - 2:   # the last layer: heads in the epilogue, no output tensor
                hw, hb = C.c_void_p(self.head_w32.data_ptr()), C.c_void_p(self.head_b32.data_ptr())
                for _, s, xp, yp, n_pixels, b0 in chains:
                    hp = C.c_void_p(heads[0].data_ptr() + b0 * heads[0].stride(0) * 2)
                    hv = C.c_void_p(heads[1].data_ptr() + b0 * heads[1].stride(0) * 2)
                    _lib.check(L.ccz_conv3x3_c256_heads_f16(s, yp, w2, b2_, xp, hw, hb, hp, hv, n_pixels, 1 | v2, None, 0, 1))
                continue
            for _, s, xp, yp, n_pixels, _b0 in chains:
                _lib.check(L.ccz_conv3x3_c256_f16(s, yp, w2, b2_, xp, xp, n_pixels, 1 | v2))  # output written over the residual input
        for st, *_ in chains[1:]:  # every side stream is joined into the current stream
            join = torch.cuda.Event()
            join.record(st)
            cur.wait_event(join)

    def _persistent_flag(self) -> int:
        """A/B switch ``persistent`` (``CCZ_CONV_PERSISTENT=N``): flag bits that run a group-of-16 tower layer on N persistent workgroups."""
        from . import _lib
        n = self.opt.persistent
        return (_lib.CONV_G16_PERSISTENT | ((n & 0xfff) << 16)) if n > 0 else 0

    def _force_flag(self) -> int:
        """A/B switch ``force`` (``CCZ_CONV_FORCE=small|tile``): run every convolution on k_conv3x3_small / on the 256-pixel tile
        kernel whatever the batch size (the library picks by batch size otherwise; the results are bit-identical either way)."""
        return self.opt.force

    def _stem_fused(self, leaf_input, plan=None, g16=None):
        """Stem on the same MFMA kernel: pack the 21 live planes as NHWC rows of 64 channels, then one 64-channel chunk of
        the tower convolution (conv3x3 + bias + ReLU). Replaces cat + layout copy + MIOpen convolution + epilogue pass.
        With a ``plan`` the pack GATHERS: output row i is board rows[i], for the live rows only."""
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        B = leaf_input.shape[0]
        if g16 is None:
            g16 = self._g16(B)
        s = C.c_void_p(torch.cuda.current_stream(leaf_input.device).cuda_stream)
        Bp = -(-B // 16) * 16 if g16 else B          # group-of-16 layout: whole groups; the padding boards hold zeros
        lay = _lib.CONV_G16 if g16 else 0   # (the stem is one launch over the whole batch, not chained: the single-kernel form)
        if plan is None:
            x64 = (torch.zeros if Bp != B else torch.empty)((Bp, 90, 64), dtype=torch.float16, device=leaf_input.device)
            y = torch.empty((Bp, 256, 10, 9), dtype=torch.float16, device=leaf_input.device, memory_format=torch.channels_last)
            if g16:
                _lib.check(L.ccz_pack_live_planes_g16_f16(s, C.c_void_p(leaf_input.data_ptr()), C.c_void_p(x64.data_ptr()), B, None, None))
            else:
                _lib.check(L.ccz_pack_live_planes_f16(s, C.c_void_p(leaf_input.data_ptr()), C.c_void_p(x64.data_ptr()), B))
            _lib.check(L.ccz_conv3x3_stem_f16(s, C.c_void_p(x64.data_ptr()), C.c_void_p((self.stem_w64_g16 if g16 else self.stem_w64).data_ptr()), C.c_void_p(self.stem_b32.data_ptr()),
                                              C.c_void_p(y.data_ptr()), Bp * 90, 1 | self._force_flag() | lay))
            return y
        rows, n_rows = plan
        # rows past the live ones are never computed: they must still hold finite numbers for the heads' GEMMs (whose results
        # for those rows nobody reads). Two persistent buffers, zeroed ONCE: whatever a row holds later is an old finite result.
        bufs = self.__dict__.setdefault("_plan_bufs", {})
        key = (Bp, leaf_input.device, g16)
        if key not in bufs:
            bufs[key] = (torch.zeros((Bp, 90, 64), dtype=torch.float16, device=leaf_input.device),
                         torch.empty((Bp, 256, 10, 9), dtype=torch.float16, device=leaf_input.device, memory_format=torch.channels_last).zero_())
        x64, y = bufs[key]
        pack = L.ccz_pack_live_planes_g16_f16 if g16 else L.ccz_pack_live_planes_rows_f16
        _lib.check(pack(s, C.c_void_p(leaf_input.data_ptr()), C.c_void_p(x64.data_ptr()), B, C.c_void_p(rows.data_ptr()), C.c_void_p(n_rows.data_ptr())))
        _lib.check(L.ccz_conv3x3_stem_f16_live(s, C.c_void_p(x64.data_ptr()), C.c_void_p((self.stem_w64_g16 if g16 else self.stem_w64).data_ptr()), C.c_void_p(self.stem_b32.data_ptr()),
                                               C.c_void_p(y.data_ptr()), Bp * 90 if g16 else -(-B // 8) * 8 * 90, 1 | lay, C.c_void_p(n_rows.data_ptr()), 0, 1))
        return y

    @torch.no_grad()
    def tower_activations(self, leaf_input):
        """Stem + tower output as a [B, 256, 10, 9] channels-last tensor in BOARD order, whatever row layout the kernels ran in
        (tests and probes; the evaluator itself feeds the heads in memory order)."""
        B = leaf_input.shape[0]
        g16 = self._g16(B)
        x = self._tower_fused(self._stem_fused(leaf_input, None, g16), None, g16)
        if not g16:
            return x
        Bp, Cn = x.shape[0], x.shape[1]
        rows = x.permute(0, 2, 3, 1).reshape(Bp // 16, 90, 16, Cn).permute(0, 2, 1, 3).reshape(Bp, 10, 9, Cn)[:B]
        return rows.permute(0, 3, 1, 2)

    def _path(self, leaf_input) -> str:
        """ONE decision per batch shape (cached; ``set_options`` clears the cache): which kernels the stem and the tower run on.
        "g16" / "nhwc": pack + stem + tower on the hand-written MFMA kernels, rows in the group-of-16 / board-major order;
        "torch_stem": the stem through torch (full 119-plane input or ``fused_stem`` off), the tower on the hand-written kernels
        when it is 256 wide fp16 on the GPU; "torch": everything through torch / MIOpen + the one-pass epilogue."""
        B = leaf_input.shape[0]
        key = (B, leaf_input.is_cuda, leaf_input.dtype, leaf_input.is_contiguous())
        cache = self.__dict__.setdefault("_path_cache", {})
        if key not in cache:
            o = self.opt
            fused = (hasattr(self, "stem_w64") and leaf_input.is_cuda and leaf_input.dtype == torch.float16 and leaf_input.is_contiguous()
                     and B >= self.FUSED_MIN_BOARDS and self.dtype == torch.float16 and o.fused_conv and o.fused_stem)
            if fused:
                cache[key] = "g16" if self._g16(B) else "nhwc"
            else:
                cache[key] = "torch_stem" if (o.fused_conv and leaf_input.is_cuda and self.dtype == torch.float16
                                              and self.ws and self.ws[0].shape[0] == 256 and self.ws[0].shape[1] == 256) else "torch"
        return cache[key]

    @torch.no_grad()
    def forward(self, leaf_input: torch.Tensor, return_logits: bool = False, plan=None):
        """``plan`` = (rows int32 [B], n_rows int32 [1]) device tensors of the planned evaluator boundary (``ccz_eval_plan``):
        outputs are COMPACT -- row i is the evaluation of board rows[i], for i < n_rows; the other rows are unspecified. On the
        fused path only the live rows are computed; elsewhere the rows are gathered and the whole batch is evaluated."""
        B = leaf_input.shape[0]
        path = self._path(leaf_input)
        if plan is not None and path not in ("g16", "nhwc"):
            leaf_input = leaf_input.index_select(0, plan[0].long().clamp_(0, B - 1))
            plan = None
        g16 = path == "g16"   # True: x holds its rows in the group-of-16 layout (and ceil(B / 16) * 16 boards)
        if path in ("g16", "nhwc"):
            x = self._stem_fused(leaf_input, plan, g16)
            probe = self.__dict__.get("tower_probe")   # bench.py: a list that receives one HIP-event pair around the 80 tower launches
            if probe is not None:   # (recorded on the current stream: the chains fork from it and join it)
                p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                p0.record()
            # group-of-16 rows: the head convolutions ride in the last tower layer's epilogue (fused_last; CCZ_FUSED_LAST=0: a pass of their own)
            heads = self._head_buffers(x.shape[0], x.device)[:2] if (g16 and self.opt.fused_last and self._fused_heads_ok(x)) else None
            x = self._tower_fused(x, plan, g16, heads)
            if probe is not None:
                p1.record()
                probe.append((p0, p1))
            if heads is not None:
                return self._heads(x, B, g16, plan, return_logits, conv_done=True)
        else:
            x = leaf_input.view(B, PLAYS * PIECES, 10, 9)
            if self.live_only:
                x = torch.cat([x[:, 49:56], x[:, 105:119]], dim=1)
            x = x.to(self.dtype).contiguous(memory_format=torch.channels_last)
            x = self._epilogue(F.conv2d(x, self.stem_w, None, padding=1), self.stem_b)
            if path == "torch_stem" and self._use_fused_tower(x):
                x = self._tower_fused(x, None, False)
            else:
                if x.is_cuda and x.dtype == torch.float16 and self.opt.fused_conv:
                    key = (int(x.shape[0]), int(x.shape[1]))
                    seen = self.__dict__.setdefault("_off_fused_seen", set())
                    if key not in seen:  # once per shape: the caller should know this batch does not run on the MFMA kernel
                        seen.add(key)
                        from .tools import log
                        log(f"evaluator: batch of {key[0]} boards x {key[1]} channels is off the fused tower kernels "
                            f"(k_conv3x3_g16 / k_conv3x3_small serve 256-channel towers): MIOpen convolutions + one-pass epilogue")
                for i in range(0, len(self.ws), 2):
                    y = self._epilogue(F.conv2d(x, self.ws[i], None, padding=1), self.bs[i])
                    x = self._epilogue(F.conv2d(y, self.ws[i + 1], None, padding=1), self.bs[i + 1], x)
        return self._heads(x, B, g16, plan, return_logits)

    def _fused_heads_ok(self, x) -> bool:
        return bool(self.opt.fused_heads and hasattr(self, "head_w32") and x.is_cuda and x.dtype == torch.float16 and x.shape[1] == 256
                    and x.is_contiguous(memory_format=torch.channels_last))

    def _head_buffers(self, Bx, dev):
        """(pol [Bx, 1536], val [Bx, 640], h1 [Bx, 256]) fp16: the head convolutions' outputs in board order and the hidden value
        layer. The pad columns of pol / val are never written and must be zero: allocated once per batch shape, zeroed once."""
        from . import _lib
        bufs = self.__dict__.setdefault("_head_bufs", {})
        key = (Bx, dev)
        if key not in bufs:
            bufs[key] = (torch.zeros((Bx, _lib.HEAD_POL_STRIDE), dtype=torch.float16, device=dev),
                         torch.zeros((Bx, _lib.HEAD_VAL_STRIDE), dtype=torch.float16, device=dev),
                         torch.empty((Bx, 256), dtype=torch.float16, device=dev))
        return bufs[key]

    def _heads_fused(self, x, B, g16, plan, conv_done=False):
        """The evaluator's tail on the hand-written kernels (csrc/cczero_heads.h): both 1x1 head convolutions + ReLU + the
        group-of-16 -> board permutation in ONE pass over the tower's rows, the two big FC layers as MFMA GEMMs, value_fc2 + tanh
        -- on the LIVE rows only (``plan``: a device-side count), the same bits for a board at every batch size. Returns
        (logits fp16 [B, 2086], value float32 [B])."""
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        Bx, dev = x.shape[0], x.device
        pol, val, h1 = self._head_buffers(Bx, dev)
        logits = torch.empty((B, 2086), dtype=torch.float16, device=dev)
        v = torch.empty((B,), dtype=torch.float32, device=dev)
        s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        live = C.c_void_p(plan[1].data_ptr()) if plan is not None else None
        P = lambda t: C.c_void_p(t.data_ptr())
        if not conv_done:   # (conv_done: the last tower layer wrote pol / val from its epilogue, ``x`` is not the tower's output)
            _lib.check(L.ccz_heads_conv1x1_f16(s, P(x), P(self.head_w32), P(self.head_b32), P(pol), P(val), Bx, _lib.CONV_G16 if g16 else 0, live))
        _lib.check(L.ccz_fc_f16(s, P(pol), _lib.HEAD_POL_STRIDE, P(self.policy_fc_wp), P(self.policy_fc_b32), P(logits), 2086, B, 2086, 1536, 0, live))
        _lib.check(L.ccz_fc_f16(s, P(val), _lib.HEAD_VAL_STRIDE, P(self.value_fc1_wp), P(self.value_fc1_b32), P(h1), 256, B, 256, 640, 1, live))
        _lib.check(L.ccz_value_out_f32(s, P(h1), P(self.value_fc2_w), self._value_b2(), P(v), B, live))
        return logits, v

    def _value_b2(self) -> float:
        """value_fc2's bias as the host float the kernel takes by value (read once; ``repack_derived`` refreshes it)."""
        b2 = self.__dict__.get("_b2")
        if b2 is None:
            b2 = self.__dict__["_b2"] = float(self.value_fc2_b.detach().float().cpu().item())
        return b2

    def _heads(self, x, B, g16, plan, return_logits, conv_done=False):
        """Policy / value heads and FC layers (reference net.py:96-109) on the tower's output ``x`` (channels-last rows in memory
        order; ``g16``: group-of-16 row order, padded to whole groups)."""
        if self._fused_heads_ok(x):
            logits, v = self._heads_fused(x, B, g16, plan, conv_done)
            if return_logits:
                return logits, v
            return torch.exp(F.log_softmax(logits.float(), dim=1)).contiguous(), v
        Bx = x.shape[0]
        rows = x.permute(0, 2, 3, 1).reshape(Bx * 90, x.shape[1])   # a view of the activations in memory order: one row per pixel
        h = F.relu_(torch.addmm(self.head_b, rows, self.head_wT))
        if g16:  # rows (g * 90 + pos) * 16 + j -> board 16 g + j, pixel pos (a copy of B x 90 x 24 values)
            h = h.view(Bx // 16, 90, 16, PLAYS + PIECES).permute(0, 2, 1, 3).reshape(Bx, 90, PLAYS + PIECES)[:B]
        else:
            h = h.view(B, 90, PLAYS + PIECES)
        pol = h[:, :, :PLAYS].reshape(B, 90 * PLAYS)       # (pixel, channel) order: the FC weights' columns are permuted to match
        val = h[:, :, PLAYS:].reshape(B, 90 * PIECES)
        logits = F.linear(pol, self.policy_fc_w, self.policy_fc_b)
        v = F.relu_(F.linear(val, self.value_fc1_w, self.value_fc1_b))
        v = torch.tanh(F.linear(v, self.value_fc2_w, self.value_fc2_b).float()).view(B)
        if return_logits:  # compact boundary: the engine applies exp(log_softmax) to the legal ids only
            return logits.contiguous(), v.contiguous()
        prob = torch.exp(F.log_softmax(logits.float(), dim=1))
        return prob.contiguous(), v.contiguous()


class PolicyValueNet:
    """Mirror of reference net.py:113-247 (same constructor, ``policy_value``, ``policy_value_fn``,
    ``save_model``, ``train_step``) plus the batched evaluator :meth:`evaluate_leaves`."""

    def __init__(self, model=None, use_gpu=True, device=None, num_channels=256, resblocks_num=40):
        self.use_gpu = use_gpu
        self.l2_const = 2e-3
        if device is not None:
            self.device = torch.device(device)
        else:
            self.device = torch.device("cuda") if (use_gpu and torch.cuda.is_available()) else torch.device("cpu")
        self.policy_value_net = Net(num_channels, resblocks_num).to(self.device)
        self.optimizer = torch.optim.Adam(params=self.policy_value_net.parameters(), lr=1e-3, betas=(0.9, 0.999),
                                          eps=1e-8, weight_decay=self.l2_const)
        if model:
            self.policy_value_net.load_state_dict(torch.load(model, map_location=self.device))
            # the planes a trained net expects follow cchess's PIECE_TYPES numbering, which cannot be read in this image:
            # say which rule preset the encoders use (never switched silently)
            from . import tools
            tools.log(f"weights loaded from {model}: piece planes and legal-move order follow rule preset '{tools.PRESET}' "
                      "(reference-trained weights: try tools.set_rules(preset='python-chess-lineage') [unverified] if this one plays nonsense)")
        self._infer = None
        self._graph = None
        # MIOpen "find" mode for the evaluator's torch convolutions (towers that are not 256 wide): read once; CCZ_MIOPEN_FIND=0 =
        # immediate mode, no per-shape find step (tests, tiny nets)
        self._miopen_find = os.environ.get("CCZ_MIOPEN_FIND", "1") != "0"
        self.weights_version = 0  # bumped whenever the inference copy is rebuilt or invalidated (hipGraphs hold its addresses)

    def invalidate_inference_copy(self):
        """The training weights changed: the fp16 inference copy is stale (rebuilt on the next evaluation)."""
        self._infer = None
        self.weights_version += 1

    # ---- batched evaluator for the lockstep engine -------------------------------------------
    def refresh_inference_copy(self):
        # MIOpen "find" mode: PyTorch's default immediate mode picks the asm implicit-GEMM kernel for the
        # [B,256,10,9] 3x3 convolutions (0.87 ms at B=4096); the find step measures all applicable solvers once
        # per shape and selects the composable-kernel XDL grouped-conv kernel (0.60 ms) -- profiles/miopen_find_r01.txt
        self._require_current_fp32("refresh_inference_copy")
        self.policy_value_net.eval()
        self._infer = InferenceNet(self.policy_value_net).to(self.device).eval()
        self._infer.bind_chain_streams(self.device)   # hardware queues for the tower's launch chains, before anything else asks for streams
        self._graph = None
        self.weights_version += 1
        return self._infer

    def _require_current_fp32(self, what: str):
        """After ``replay.broadcast_model(what="inference")`` a receiving rank's fp32 ``Net`` still holds its OLD weights (only the
        fp16 inference copy was overwritten): rebuilding the inference copy from it, saving it or training it would silently revert
        the reload. Such a rank only evaluates; anything else needs ``broadcast_model(what="state")`` first."""
        if getattr(self, "_fp32_stale", False):
            raise RuntimeError(f"{what}: this rank's fp32 net does not hold the weights of the last broadcast_model(what='inference') "
                               "(only the inference copy was sent): call broadcast_model(what='state') before training, saving or "
                               "rebuilding the inference copy here")

    @torch.no_grad()
    def evaluate_leaves(self, leaf_input: torch.Tensor):
        """[B,17,7,10,9] fp16 device tensor -> (prob float32 [B,2086], value float32 [B]) on the same stream."""
        if self._infer is None:
            self.refresh_inference_copy()
        if not self._miopen_find:
            return self._infer(leaf_input)
        with torch.backends.cudnn.flags(enabled=True, benchmark=True):  # find mode for the evaluator's convs only
            return self._infer(leaf_input)

    evaluate_leaves.batched = True
    evaluate_leaves.graph_safe = True   # static shapes, no host sync: may be captured into a hipGraph

    @torch.no_grad()
    def evaluate_leaves_logits(self, leaf_input: torch.Tensor, plan=None):
        """Same as :meth:`evaluate_leaves` but returns the policy head's logits ([B,2086], fp16 on the GPU): the
        engine's ``ccz_step_logits`` turns them into priors of the legal moves only. ``plan``: the planned boundary of an engine
        with an evaluation cache (``SelfPlayEngine.eval_plan()``): only the planned rows are computed, outputs are compact."""
        if self._infer is None:
            self.refresh_inference_copy()
        if not self._miopen_find:
            return self._infer(leaf_input, return_logits=True, plan=plan)
        with torch.backends.cudnn.flags(enabled=True, benchmark=True):
            return self._infer(leaf_input, return_logits=True, plan=plan)

    evaluate_leaves_logits.batched = True
    evaluate_leaves_logits.graph_safe = True
    evaluate_leaves_logits.returns_logits = True
    evaluate_leaves_logits.accepts_plan = True

    # ---- reference surface --------------------------------------------------------------------
    def policy_value(self, state_batch):
        """net.py:137-148"""
        self.policy_value_net.eval()
        if isinstance(state_batch, torch.Tensor):
            state_batch = state_batch.to(self.device)
        else:
            state_batch = torch.tensor(np.asarray(state_batch), dtype=torch.float).to(self.device)
        with torch.no_grad():
            log_act_probs, value = self.policy_value_net(state_batch.float())
        return np.exp(log_act_probs.cpu().numpy()), value.cpu().numpy()

    def policy_value_fn(self, board, red_states=None, black_states=None):
        """Single-board evaluator with the reference's signature and return convention (net.py:151-205):
        ``(zip(legal ids, P[ids]), value ndarray(1,1))``. ``board`` is a :class:`game.Board`."""
        from .tools import decode_board
        self.policy_value_net.eval()
        legal_positions = board.legal_ids()
        if red_states is None or black_states is None:
            red_state, black_state = decode_board(board)
            red_states = [np.zeros((PIECES, 10, 9), dtype=np.float16) for _ in range(PIECES)] + [red_state]
            black_states = [np.zeros((PIECES, 10, 9), dtype=np.float16) for _ in range(PIECES)] + [black_state]
        current_player = (np.ones if board.turn else np.zeros)((1, PIECES, 10, 9), dtype=np.float16)
        states = np.concatenate((red_states, black_states, current_player), axis=0)
        x = torch.as_tensor(np.ascontiguousarray(states.reshape(-1, PLAYS, PIECES, 10, 9)).astype("float16"))
        with torch.no_grad():
            if self.device.type == "cuda":
                with torch.autocast("cuda"):
                    log_act_probs, value = self.policy_value_net(x.to(self.device))
            else:
                log_act_probs, value = self.policy_value_net(x.float().to(self.device))
        act_probs = np.exp(log_act_probs.float().cpu().numpy().flatten())
        return zip(legal_positions, act_probs[legal_positions]), value.float().cpu().numpy()

    def save_model(self, model_file):
        self._require_current_fp32("save_model")
        torch.save(self.policy_value_net.state_dict(), model_file)

    def train_step(self, state_batch, mcts_probs, winner_batch, lr=0.002):
        """net.py:212-247 (plain PyTorch; the trainer is a consumer of the rollout path, not part of it)."""
        self._require_current_fp32("train_step")
        self.policy_value_net.train()
        to = lambda t: (t if isinstance(t, torch.Tensor) else torch.as_tensor(np.asarray(t))).to(self.device, dtype=torch.float)
        state_batch, mcts_probs, winner_batch = to(state_batch), to(mcts_probs), to(winner_batch)
        self.optimizer.zero_grad()
        log_act_probs, value = self.policy_value_net(state_batch)
        value_loss = F.mse_loss(input=torch.reshape(value, shape=[-1]), target=winner_batch)
        policy_loss = -torch.mean(torch.sum(mcts_probs * log_act_probs, dim=1))
        loss = value_loss + policy_loss
        loss.backward()
        self.optimizer.step()
        with torch.no_grad():
            entropy = -torch.mean(torch.sum(torch.exp(log_act_probs) * log_act_probs, dim=1))
        self.invalidate_inference_copy()
        return loss.detach().cpu().numpy(), entropy.detach().cpu().numpy()


def uniform_evaluator(leaf_input: torch.Tensor):
    """Constant-time stub evaluator (uniform priors, value 0) used to time the simulator alone."""
    B = leaf_input.shape[0]
    key = (B, leaf_input.device)
    cache = uniform_evaluator.__dict__.setdefault("_cache", {})
    if key not in cache:
        cache[key] = (torch.full((B, 2086), 1.0 / 2086, dtype=torch.float32, device=leaf_input.device),
                      torch.zeros((B,), dtype=torch.float32, device=leaf_input.device))
    return cache[key]


uniform_evaluator.batched = True
