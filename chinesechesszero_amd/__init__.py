"""chinesechesszero_amd -- MI355X-native self-play rollout engine for ChineseChessZero.

Only what the hot path needs: the gfx950 HIP engine behind a C ABI
(``csrc/``, ``include/cczero.h``) and the host-side mirror of the reference's call surface
(``tools``, ``parameters``, ``net``, ``mcts``, ``game``, ``collect``) plus the tuple all-gather
(``replay``). There is no CPU fallback: the engine fails loudly when ``libcczero.so`` or a GPU is
missing.
"""
import os as _os

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4). The evaluator runs its tower as three concurrent launch chains
# on streams of its own; as soon as a process also holds the streams of the multi-GPU exchange (the process group's RCCL stream, the
# exchange's side stream), four queues are not enough: chains end up sharing a queue and run one after the other -- measured on one
# MI355X with the N>1 path in a group of one on RCCL: 26.6 ms per step instead of 21.2 (tower layer 330 us instead of 262), with 8
# queues 21.1 (profiles/r05_h_*.json, r05_g1_*.json). The variable is read when the HIP runtime initialises, i.e. at the first device
# call, not at ``import torch``: importing this package first is early enough. A value set by the user wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
try:
    if int(_os.environ["GPU_MAX_HW_QUEUES"]) < 8:
        import warnings as _warnings
        _warnings.warn(f"GPU_MAX_HW_QUEUES={_os.environ['GPU_MAX_HW_QUEUES']} (set by the caller): with fewer than 8 hardware queues the evaluator's "
                       "launch chains can share queues with the streams of a multi-GPU process group and run ~20 % slower (DESIGN.md section 7)")
except ValueError:
    pass

__version__ = "0.1.0"
