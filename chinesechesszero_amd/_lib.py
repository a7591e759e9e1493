"""ctypes binding of libcczero.so (C ABI: include/cczero.h). No fallback: import fails loudly.

``torch`` is imported first on purpose: PyTorch-ROCm ships its own HIP runtime with the same
soname as the system one; loading it first makes libcczero.so share that single runtime, so
streams and device pointers can be passed across the boundary.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL load, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcczero.so")

NSQ = 90
SQ_STRIDE = 96
NMOVES = 2086
MAX_LEGAL = 128
MASK_WORDS = 66
PLANES = 10710
REC_BYTES, REC_HDR, REC_IDS, REC_PI = 880, 96, 112, 368  # compact ply record (include/cczero.h CCZ_REC_*)
HEAD_POL_STRIDE, HEAD_VAL_STRIDE = 1536, 640  # CCZ_HEAD_*_STRIDE: fp16 elements per board of the head kernels' outputs

ABI_VERSION = 7
CONV_RELU, CONV_DESCENDING, CONV_FORCE_SMALL, CONV_FORCE_TILE, CONV_G16, CONV_G16_EDGE_TILES, CONV_G16_PERSISTENT = 1, 2, 16, 32, 64, 128, 256  # CCZ_CONV_* flag bits
RULE_PERPETUAL_CHECK = 1
RULE_PAWN_MOVE_RESETS_CLOCK = 2
FLAG_REFERENCE_QUIRKS = 1
FLAG_NO_MIRROR = 2
FLAG_VALUE_F16 = 4
FLAG_CACHE_VERIFY = 8
ERR_PRUNED, ERR_TRUNCATED = 256, 512   # CCZ_ERR_* set in strict mode only
FLAG_STRICT = 16   # parity mode: pruning a kept subtree / adjudicating at max_plies are error bits, not counters
LEAF_EXPAND, LEAF_DRAW, LEAF_LOSS, LEAF_SKIP = 0, 1, 2, 3

ERR_BITS = {1: "node pool exhausted (raise max_nodes)", 2: "selection path deeper than max_depth", 64: "history chain overflow (> 128 positions since the last capture)",
            4: "more than 128 legal moves or pseudo-move overflow", 8: "pi record arena overflow",
            16: "forced move is not a child of the root / root not expanded", 32: "NaN priors",
            128: "index out of bounds (bounds-checked diagnostic build)",
            256: "strict mode: a kept subtree was pruned to fit the node pool (the reference's tree is unbounded: raise max_nodes)",
            512: "strict mode: a game was adjudicated at max_plies (the reference's game has no ply cap: raise max_plies)"}


class CczError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [
        ("n_boards", C.c_int32), ("n_playout", C.c_int32), ("c_puct", C.c_float), ("eps", C.c_float),
        ("alpha", C.c_float), ("temp", C.c_float), ("max_nodes", C.c_int32), ("max_depth", C.c_int32),
        ("max_plies", C.c_int32), ("flags", C.c_uint32), ("seed", C.c_uint64), ("board_id_base", C.c_uint64),
        ("device", C.c_int32), ("reserve_nodes", C.c_int32),
        ("move_rank_host", C.c_void_p), ("plane_of_type", C.c_uint8 * 8), ("rule_flags", C.c_uint32), ("eval_cache_log2", C.c_uint32),
        ("type_rank", C.c_uint8 * 8),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("sims", C.c_int64), ("moves", C.c_int64), ("games", C.c_int64), ("truncated_games", C.c_int64),
        ("nodes_peak", C.c_int64), ("depth_peak", C.c_int64), ("sum_depth", C.c_int64), ("sum_children", C.c_int64),
        ("expansions", C.c_int64), ("terminal_leaves", C.c_int64), ("error_flags", C.c_int32), ("reserved", C.c_int32),
        ("hbm_bytes", C.c_int64), ("pruned_subtrees", C.c_int64),
        ("cache_probes", C.c_int64), ("cache_hits", C.c_int64), ("cache_shared_rows", C.c_int64), ("cache_stores", C.c_int64),
        ("cache_verified", C.c_int64), ("cache_verify_mismatches", C.c_int64),
    ]


# every symbol include/cczero.h declares: name -> (restype, argtypes)
_P = C.c_void_p
PROTOTYPES = {
    "ccz_abi_version": (C.c_int, []),
    "ccz_last_error": (C.c_char_p, []),
    "ccz_device_count": (C.c_int, []),
    "ccz_action_table": (C.c_int, [_P, _P, _P]),
    "ccz_flip_map": (C.c_int, [_P]),
    "ccz_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "ccz_destroy": (C.c_int, [_P]),
    "ccz_reset": (C.c_int, [_P, _P, _P]),
    "ccz_set_position": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, C.c_int32]),
    "ccz_reset_tree": (C.c_int, [_P, _P, _P]),
    "ccz_select_leaves": (C.c_int, [_P, _P, _P]),
    "ccz_zero_leaf_input": (C.c_int, [_P, _P, _P]),
    "ccz_expand_backup": (C.c_int, [_P, _P, _P, _P]),
    "ccz_step": (C.c_int, [_P, _P, _P, _P, _P]),
    "ccz_gather_priors": (C.c_int, [_P, _P, _P, C.c_int32]),
    "ccz_step_compact": (C.c_int, [_P, _P, _P, _P]),
    "ccz_expand_backup_compact": (C.c_int, [_P, _P, _P]),
    "ccz_eval_plan": (C.c_int, [_P, _P, _P, _P]),
    "ccz_set_scouts": (C.c_int, [_P, C.c_int32]),
    "ccz_scout": (C.c_int, [_P, _P, _P]),
    "ccz_eval_plan_scouted": (C.c_int, [_P, _P, _P, _P, _P]),
    "ccz_scout_and_plan": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "ccz_scouted_run": (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "ccz_gather_priors_planned": (C.c_int, [_P, _P, _P, C.c_int32, _P]),
    "ccz_eval_cache_clear": (C.c_int, [_P, _P]),
    "ccz_finish_move": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32]),
    "ccz_root_children": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "ccz_root_pi": (C.c_int, [_P, _P, _P, _P]),
    "ccz_game_status": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "ccz_root_positions": (C.c_int, [_P, _P, _P]),
    "ccz_leaf_info": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "ccz_leaf_priors": (C.c_int, [_P, _P, _P, _P]),
    "ccz_leaf_keys": (C.c_int, [_P, _P, _P, _P]),
    "ccz_harvest_rows": (C.c_int, [_P, _P, C.POINTER(C.c_int64)]),
    "ccz_harvest": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "ccz_harvest_records": (C.c_int, [_P, _P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "ccz_expand_records": (C.c_int, [_P, _P, C.c_int64, C.c_uint32, _P, _P, _P, _P, C.c_int64, C.c_int64, _P]),
    "ccz_get_stats": (C.c_int, [_P, _P, C.POINTER(Stats)]),
    "ccz_legal_moves": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "ccz_apply_moves": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P]),
    "ccz_bias_act_f16": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32]),
    "ccz_conv3x3_c256_f16": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, C.c_int32]),
    "ccz_conv3x3_stem_f16": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_int32]),
    "ccz_pack_live_planes_f16": (C.c_int, [_P, _P, _P, C.c_int32]),
    "ccz_pack_live_planes_rows_f16": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P]),
    "ccz_pack_live_planes_g16_f16": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P]),
    "ccz_pack_conv_weights_g16_f16": (C.c_int, [_P, _P, _P, C.c_int32]),
    "ccz_conv3x3_c256_f16_live": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, C.c_int32, _P, C.c_int32, C.c_int32]),
    "ccz_conv3x3_stem_f16_live": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_int32, _P, C.c_int32, C.c_int32]),
    "ccz_conv3x3_c256_heads_f16": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int32, _P, C.c_int32, C.c_int32]),
    "ccz_heads_conv1x1_f16": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P]),
    "ccz_fc_f16": (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "ccz_value_out_f32": (C.c_int, [_P, _P, _P, C.c_float, _P, C.c_int32, _P]),
}

_lib = None


def source_hash() -> str | None:
    """16 hex digits over the library's sources as they lie in the tree (the csrc Makefile writes the same digest next to the library when
    it builds it: ``libcczero.so.srchash``); None where the sources are not there (an installed copy)."""
    import glob
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.hip")), key=os.path.basename)
    files.append(os.path.join(os.path.dirname(_HERE), "include", "cczero.h"))
    if len(files) < 2 or not all(os.path.exists(f) for f in files):
        return None
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def stale_build() -> str | None:
    """Why the in-tree libcczero.so must not be used (it was built from other sources than the ones in the tree), or None."""
    if LIB_PATH != os.path.join(_HERE, "libcczero.so"):
        return None                                  # a diagnostic build chosen with CCZ_LIB: its own business
    want = source_hash()
    if want is None:
        return None
    stamp = LIB_PATH + ".srchash"
    if not os.path.exists(stamp):
        return f"{LIB_PATH} carries no source stamp ({stamp} is missing): it was not built by the current Makefile"
    with open(stamp) as f:
        got = f.read().strip()
    if got != want:
        return f"{LIB_PATH} was built from other sources (stamp {got}, the tree hashes to {want}): a kernel was edited after the last build"
    return None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CczError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). The engine has no CPU fallback.")
        why = stale_build()
        if why:
            raise CczError(why + " -- rebuild with `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if L.ccz_abi_version() != ABI_VERSION:
            raise CczError("libcczero.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise CczError(f"libcczero error {rc}: {lib().ccz_last_error().decode(errors='replace')}")
