"""chinesechesszero_amd -- MI355X-native self-play rollout engine for ChineseChessZero.

Only what the hot path needs: the gfx950 HIP engine behind a C ABI
(``csrc/``, ``include/cczero.h``) and the host-side mirror of the reference's call surface
(``tools``, ``parameters``, ``net``, ``mcts``, ``game``, ``collect``) plus the tuple all-gather
(``replay``). There is no CPU fallback: the engine fails loudly when ``libcczero.so`` or a GPU is
missing.
"""
__version__ = "0.1.0"
