"""CPU: the trainer-format tuple sink against the file format the reference's converter writes
(convert.py:84-99) and its dataset reads (dataset.py:45-89). No GPU needed: rows are synthetic."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rows(n, seed):
    rs = np.random.RandomState(seed)
    s = (rs.rand(n, 17, 7, 10, 9) > 0.9).astype(np.float16)
    p = rs.rand(n, 2086)
    p /= p.sum(1, keepdims=True)
    z = rs.choice([-1.0, 0.0, 1.0], size=n).astype(np.float32)
    return s, p, z


def test_meta_json_is_the_converters(tmp_path):
    """tests/golden/meta_reference_format.json is hand-derived from convert.py:89-99 for 160 float64 rows: same keys, same
    order, same values, same text layout (json.dump(..., ensure_ascii=False, indent=2)); `iters` is this build's one extra key."""
    from chinesechesszero_amd.collect import TupleSink
    sink = TupleSink(str(tmp_path))
    s, p, z = _rows(160, 0)
    sink.append(s[:100], p[:100], z[:100], games=3)
    sink.append(s[100:], p[100:], z[100:], games=2)
    assert not os.path.exists(tmp_path / "states.npy") and sink.rows() == 160      # collecting never rewrites the data set
    assert sink.finalize() == 160
    golden_text = open(os.path.join(ROOT, "tests", "golden", "meta_reference_format.json"), encoding="utf-8").read()
    golden = json.loads(golden_text)
    text = open(tmp_path / "meta.json", encoding="utf-8").read()
    meta = json.loads(text)
    assert list(meta)[:len(golden)] == list(golden) and all(meta[k] == golden[k] for k in golden)
    assert list(meta)[len(golden):] == ["iters"] and meta["iters"] == 5
    assert text.startswith(golden_text.rstrip()[:-1].rstrip())                       # byte-identical up to the extra key
    assert np.load(tmp_path / "mcts.npy").dtype == np.float64 and np.array_equal(np.load(tmp_path / "mcts.npy"), p)
    assert np.array_equal(np.load(tmp_path / "states.npy"), s) and np.array_equal(np.load(tmp_path / "winners.npy"), z)
    assert not [f for f in os.listdir(tmp_path) if f.startswith(".shard_")]


def test_incremental_finalize_resume_and_dataset(tmp_path):
    from chinesechesszero_amd.collect import TupleSink
    from chinesechesszero_amd.dataset import NpyMemmapDataset
    s, p, z = _rows(50, 1)
    a = TupleSink(str(tmp_path))
    a.append(s[:20], p[:20], z[:20], games=1)
    assert a.finalize() == 20
    a.append(s[20:30], p[20:30], z[20:30], games=1)
    # a new process opens the directory: the merged rows, the game counter and the pending shard are all picked up
    b = TupleSink(str(tmp_path))
    assert b.games == 2 and b.rows() == 10
    b.append(s[30:], p[30:], z[30:], games=2)
    assert b.finalize() == 50 and b.games == 4
    ds = NpyMemmapDataset(str(tmp_path))
    assert len(ds) == 50
    st, pi, w = ds[37]
    assert np.array_equal(st.numpy(), s[37]) and np.allclose(pi.numpy(), p[37].astype(np.float32)) and float(w) == float(z[37])
    assert json.load(open(tmp_path / "meta.json"))["total_count"] == 50
    assert b.finalize() == 50                                                           # idempotent


def test_interrupted_finalize_is_repaired(tmp_path):
    """A crash between the array replacements leaves files of different lengths; meta.json (written last) still says what
    is valid, the shards are still there, and the next finalize() completes the merge."""
    from chinesechesszero_amd.collect import TupleSink
    s, p, z = _rows(30, 2)
    a = TupleSink(str(tmp_path))
    a.append(s[:10], p[:10], z[:10])
    a.finalize()
    a.append(s[10:], p[10:], z[10:])
    real_replace = os.replace
    calls = {"n": 0}

    def dying_replace(src, dst):
        calls["n"] += 1
        if calls["n"] == 3:   # 1 = the merge journal, 2 = states.npy, 3 = mcts.npy
            raise KeyboardInterrupt("killed between two replacements")
        return real_replace(src, dst)

    os.replace = dying_replace
    try:
        with pytest.raises(KeyboardInterrupt):
            a.finalize()
    finally:
        os.replace = real_replace
    assert json.load(open(tmp_path / "meta.json"))["total_count"] == 10            # still describes the valid prefix
    assert np.load(tmp_path / "states.npy", mmap_mode="r").shape[0] == 30 and np.load(tmp_path / "mcts.npy", mmap_mode="r").shape[0] == 10
    a.close()
    b = TupleSink(str(tmp_path))       # opening the directory completes the journaled merge
    assert b.rows() == 0 and b.finalize() == 30
    assert np.array_equal(np.load(tmp_path / "states.npy"), s) and np.array_equal(np.load(tmp_path / "mcts.npy"), p)
    assert np.array_equal(np.load(tmp_path / "winners.npy"), z)
    assert json.load(open(tmp_path / "meta.json"))["total_count"] == 30 and not os.path.exists(tmp_path / "merge_journal.json")
    assert not [f for f in os.listdir(tmp_path) if f.startswith(".shard_")]


def test_crash_between_meta_and_shard_deletion_does_not_duplicate_rows(tmp_path):
    """meta.json already carries the new total, the shards are still on disk. The journal names them as merged:
    the next sink deletes them instead of adopting them, so nothing is appended twice."""
    from chinesechesszero_amd.collect import TupleSink
    s, p, z = _rows(24, 5)
    a = TupleSink(str(tmp_path))
    a.append(s[:8], p[:8], z[:8])
    a.finalize()
    a.append(s[8:], p[8:], z[8:])
    real_remove = os.remove

    def dying_remove(path):
        if ".shard_" in str(path):
            raise KeyboardInterrupt("killed before the first shard was deleted")
        return real_remove(path)

    os.remove = dying_remove
    try:
        with pytest.raises(KeyboardInterrupt):
            a.finalize()
    finally:
        os.remove = real_remove
    assert json.load(open(tmp_path / "meta.json"))["total_count"] == 24 and [f for f in os.listdir(tmp_path) if f.startswith(".shard_")]
    a.close()
    b = TupleSink(str(tmp_path))
    assert b.rows() == 0 and not [f for f in os.listdir(tmp_path) if f.startswith(".shard_")]
    b.append(s[:4], p[:4], z[:4])
    assert b.finalize() == 28
    assert np.array_equal(np.load(tmp_path / "winners.npy"), np.concatenate([z, z[:4]]))


def test_row_count_comes_from_the_files_and_everything_is_validated_before_writing(tmp_path):
    from chinesechesszero_amd.collect import TupleSink
    s, p, z = _rows(12, 6)
    a = TupleSink(str(tmp_path))
    a.append(s[:6], p[:6], z[:6])
    a.finalize()
    a.close()
    # a directory whose meta.json lacks total_count (round 1 wrote "total_samples"): the arrays are the evidence, nothing is lost
    json.dump({"total_samples": 6, "iters": 1}, open(tmp_path / "meta.json", "w"))
    b = TupleSink(str(tmp_path))
    b.append(s[6:], p[6:], z[6:])
    assert b.finalize() == 12 and np.array_equal(np.load(tmp_path / "states.npy"), s)
    b.close()
    # meta.json and the arrays disagree: refuse at open
    m = json.load(open(tmp_path / "meta.json"))
    m["total_count"] = 7
    json.dump(m, open(tmp_path / "meta.json", "w"))
    with pytest.raises(ValueError, match="meta.json says 7"):
        TupleSink(str(tmp_path))
    m["total_count"] = 12
    json.dump(m, open(tmp_path / "meta.json", "w"))
    # a shard of the wrong dtype: finalize() raises BEFORE any array is replaced (states.npy keeps its 12 rows)
    c = TupleSink(str(tmp_path))
    c.append(s[:3], p[:3], z[:3])
    base = c._shards[-1][0]
    np.save(base + "_p.npy", p[:3].astype(np.float32))
    with pytest.raises(ValueError, match="float32"):
        c.finalize()
    assert np.load(tmp_path / "states.npy", mmap_mode="r").shape[0] == 12 and np.load(tmp_path / "winners.npy", mmap_mode="r").shape[0] == 12
    c.close()


def test_a_live_collector_owns_its_directory(tmp_path):
    import subprocess
    import sys as _sys
    from chinesechesszero_amd.collect import TupleSink
    other = subprocess.Popen([_sys.executable, "-c", "import time; time.sleep(60)"])
    try:
        open(tmp_path / ".collector.lock", "w").write(str(other.pid))
        with pytest.raises(RuntimeError, match="in use by collector process"):
            TupleSink(str(tmp_path))
    finally:
        other.kill()
        other.wait()
    a = TupleSink(str(tmp_path))       # the other collector is dead: its lock is taken over
    assert open(tmp_path / ".collector.lock").read() == str(os.getpid())
    a.close()
    assert not os.path.exists(tmp_path / ".collector.lock")


def test_float32_option_and_dtype_guard(tmp_path):
    from chinesechesszero_amd.collect import TupleSink
    s, p, z = _rows(8, 3)
    a = TupleSink(str(tmp_path), pi_dtype=np.float32)
    a.append(s, p, z)
    a.finalize()
    assert np.load(tmp_path / "mcts.npy").dtype == np.float32 and json.load(open(tmp_path / "meta.json"))["mcts_dtype"] == "float32"
    a.close()
    with pytest.raises(ValueError, match="float32"):   # float64 sink on a float32 data set: refuse, do not silently convert
        TupleSink(str(tmp_path))


@pytest.mark.skipif(not os.path.exists("/root/reference/dataset.py"), reason="the reference tree is only mounted in the build container")
def test_the_references_own_dataset_reads_what_the_sink_writes(tmp_path):
    """Not a fixture but the real consumer: reference dataset.py (numpy + torch only, importable as is) opens the directory
    TupleSink.finalize() produced -- the check train.py:100 would make -- and hands back the rows, also after pickling
    (DataLoader workers, dataset.py:64-73). Build container only: the reference never travels."""
    import importlib.util
    import pickle
    from chinesechesszero_amd.collect import TupleSink
    spec = importlib.util.spec_from_file_location("ref_dataset", "/root/reference/dataset.py")
    ref = importlib.util.module_from_spec(spec)
    sys_dont = __import__("sys")
    old = sys_dont.dont_write_bytecode
    sys_dont.dont_write_bytecode = True
    try:
        spec.loader.exec_module(ref)
        sys_dont.modules["ref_dataset"] = ref     # pickle looks the class up by module name
    finally:
        sys_dont.dont_write_bytecode = old
    s, p, z = _rows(40, 9)
    sink = TupleSink(str(tmp_path))
    sink.append(s[:25], p[:25], z[:25], games=1)
    sink.append(s[25:], p[25:], z[25:], games=1)
    sink.finalize()
    ds = ref.NpyMemmapDataset(str(tmp_path))
    assert len(ds) == 40
    st, pi, w = ds[31]
    assert st.dtype == np.float16 and pi.dtype == np.float64 and w.dtype == np.float32     # what collect.py:146-167 + convert.py store
    assert np.array_equal(st, s[31]) and np.array_equal(pi, p[31]) and w == z[31]
    ds2 = pickle.loads(pickle.dumps(ds))
    assert len(ds2) == 40 and np.array_equal(ds2[7][1], p[7])
    from chinesechesszero_amd.dataset import NpyMemmapDataset
    mine = NpyMemmapDataset(str(tmp_path))
    assert len(mine) == len(ds) and np.array_equal(mine[31][0].numpy(), st)


def _fake_records(lengths, seed=0):
    """Synthetic compact ply records (include/cczero.h CCZ_REC_*): whole games, valid (t, T) headers, random payload."""
    rs = np.random.RandomState(seed)
    P = sum(lengths)
    rec = rs.randint(0, 256, size=(P, 880)).astype(np.uint8)
    hdr = np.zeros((P, 4), np.uint16)
    p = 0
    for T in lengths:
        for t in range(T):
            hdr[p] = (t, T, 0, 0)
            p += 1
    rec[:, 96:104] = hdr.view(np.uint8).reshape(P, 8)
    return rec


def test_record_shards_are_what_the_batched_collector_writes(tmp_path):
    """Round 5: while it runs the batched collector writes COMPACT ply records (880 B per ply) -- 85 x fewer bytes through the host
    than the dense rows (float16 planes + float64 pi) -- and finalize() expands them with the GPU expander. Without a GPU that
    step says so loudly and changes nothing; a new sink adopts the shards (flags and plane map ride in the file name)."""
    import pytest
    import torch
    from chinesechesszero_amd._lib import CczError, FLAG_NO_MIRROR
    from chinesechesszero_amd.collect import TupleSink
    d = str(tmp_path)
    s = TupleSink(d)
    a, b = _fake_records((5, 3, 9), 1), _fake_records((4,), 2)
    s.append_records(torch.from_numpy(a), flags=0, plane_of_type=(0, 0, 6, 1, 2, 3, 4, 5), games=3)
    s.append_records(b, flags=FLAG_NO_MIRROR, games=1)
    s.append_records(a[:0], games=2)                                  # games without records (another rank's count): only counted
    assert s.rows() == 17 * 2 + 4 * 1 and s.games == 6                # mirror images double the rows unless switched off
    names = sorted(n for n in os.listdir(d) if n.startswith(".rshard_"))
    assert len(names) == 2 and names[0].endswith("_f0_p00612345.npy") and names[1].endswith(f"_f{FLAG_NO_MIRROR}_p00123456.npy")
    assert sum(os.path.getsize(os.path.join(d, n)) for n in names) < 21 * 880 + 2 * 200      # 880 B per ply + the .npy headers
    assert np.array_equal(np.load(os.path.join(d, names[0])), a)
    if not torch.cuda.is_available():
        with pytest.raises(CczError, match="GPU expander"):
            s.finalize()
        assert sorted(n for n in os.listdir(d) if n.startswith(".rshard_")) == names and not os.path.exists(os.path.join(d, "meta.json"))
    s.close()
    s2 = TupleSink(d)                                                 # the next collector adopts the shards and the game count
    assert s2.rows() == 38 and s2.games == 6 and [r[1:] for r in s2._rshards] == [(17, 0, (0, 0, 6, 1, 2, 3, 4, 5)), (4, FLAG_NO_MIRROR, (0, 0, 1, 2, 3, 4, 5, 6))]
    # dense shards left by an INTERRUPTED expansion of a record shard that is still there are not adopted (they would count twice)
    tag = names[0][len(".rshard_"):-len(".npy")]
    for sfx, arr in (("_s.npy", np.zeros((2, 17, 7, 10, 9), np.float16)), ("_p.npy", np.zeros((2, 2086), np.float64)), ("_z.npy", np.zeros(2, np.float32))):
        np.save(os.path.join(d, f".shard_r{tag}_0000{sfx}"), arr)
    s2.close()
    s3 = TupleSink(d)
    assert s3.rows() == 38 and s3._shards == []
    s3.close()


def test_a_restarted_collector_never_reuses_a_shard_tag(tmp_path):
    """an expansion that crashed after removing a record shard leaves complete dense shards `.shard_r<pid>_<seq>_..`
    whose record shard is gone. The next sink adopts them -- and, started with the SAME pid (usual in containers), must not hand their
    tag to a new record shard: the following finalize would delete them as 'leftovers of an interrupted expansion of THIS shard'."""
    from chinesechesszero_amd.collect import TupleSink
    d = str(tmp_path)
    pid = os.getpid()
    tag = f"{pid}_000000_f0_p00123456"
    for k in range(2):                                                  # dense shards of record shard 000000; the record shard itself is gone
        for sfx, arr in (("_s.npy", np.zeros((3, 17, 7, 10, 9), np.float16)), ("_p.npy", np.full((3, 2086), 1.0 / 2086)), ("_z.npy", np.ones(3, np.float32))):
            np.save(os.path.join(d, f".shard_r{tag}_{k:04d}{sfx}"), arr)
    np.save(os.path.join(d, f".shard_{pid + 1}_000004_z.npy"), np.zeros(0, np.float32))   # (an incomplete dense shard of another pid: its number counts too)
    s = TupleSink(d)
    assert s.rows() == 6 and len(s._shards) == 2 and s._next == 5      # above every number used in the directory, whatever the name form
    s.append_records(_fake_records((4,), 3), games=1)
    s.append(np.zeros((1, 17, 7, 10, 9), np.float16), np.full((1, 2086), 1.0 / 2086), np.zeros(1, np.float32), games=1)
    new = sorted(n for n in os.listdir(d) if n.startswith(".rshard_") or n.startswith(f".shard_{pid}_"))
    assert new[0].startswith(f".rshard_{pid}_000005_") and new[1].startswith(f".shard_{pid}_000006_")
    # what the next finalize treats as leftovers of an interrupted expansion of the NEW record shard: nothing that exists
    prefix = ".shard_r" + new[0][len(".rshard_"):-len(".npy")] + "_"
    assert not [n for n in os.listdir(d) if n.startswith(prefix)]
    assert TupleSink._first_free_seq([]) == 0 and TupleSink._first_free_seq(["states.npy", ".shard_7_000010_s.npy", ".rshard_7_000002_f0_p00123456.npy"]) == 11
    s.close()
