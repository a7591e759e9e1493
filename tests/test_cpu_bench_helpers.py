"""CPU: the parts of bench.py that need no GPU -- which committed profile a bench line may quote (round 5: a profile is tied to
the code it was taken with), and that the code hash moves when a kernel source does."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WL = {"boards_per_gpu": 4096, "sims_per_move": 400, "evaluator": "net", "max_plies": 200, "blocks": 40, "channels": 256,
      "preroll_plies": 200, "align": True}


def _profile(tmp_path, **kw):
    pm = {"k_step": {"avg_ns": 33000.0, "window": {"hbm_bytes_per_launch": 4.0e7}},
          "workload": {"boards_per_gpu": 4096, "sims_per_move": 400, "evaluator": "net", "max_plies": 200}, "head": "aaaabbbbccccdddd", "tag": "r05"}
    pm.update(kw)
    p = tmp_path / "pmc_summary.json"
    p.write_text(json.dumps(pm))
    return str(p)


def test_a_profile_of_other_code_or_another_workload_is_refused(tmp_path):
    import bench
    pm, why, head = bench.committed_profile(_profile(tmp_path), "aaaabbbbccccdddd", WL)
    assert pm is not None and why is None and head == "aaaabbbbccccdddd" and pm["k_step"]["avg_ns"] == 33000.0
    pm, why, head = bench.committed_profile(_profile(tmp_path), "0000111122223333", WL)           # the code changed since the profile
    assert pm is None and "other code" in why and head == "aaaabbbbccccdddd"
    pm, why, _ = bench.committed_profile(_profile(tmp_path, head=None), "aaaabbbbccccdddd", WL)   # a profile that does not say
    assert pm is None and "no head" in why
    for key, val in (("boards_per_gpu", 1024), ("sims_per_move", 800), ("evaluator", "stub"), ("blocks", 2), ("align", False)):
        pm, why, _ = bench.committed_profile(_profile(tmp_path), "aaaabbbbccccdddd", dict(WL, **{key: val}))
        assert pm is None and "another workload" in why and key in why
    pm, why, _ = bench.committed_profile(str(tmp_path / "missing.json"), "x", WL)
    assert pm is None and "no committed profile" in why


def test_the_committed_profile_names_its_head_and_holds_no_kernel_that_no_longer_runs():
    """profiles/pmc_summary.json as committed: it says which code it was taken with, and every kernel block in it belongs to a kernel
    the profiled run launched (round 4 shipped a k_head_conv1x1 block of a kernel that had left the timed path)."""
    with open(os.path.join(ROOT, "profiles", "pmc_summary.json")) as f:
        pm = json.load(f)
    assert isinstance(pm.get("head"), str) and len(pm["head"]) == 16 and pm.get("tag")
    stats = os.path.join(ROOT, "profiles", f"{pm['tag']}_kernel_stats.csv")
    assert os.path.exists(stats)
    names = open(stats).read()
    for k, v in pm.items():
        if isinstance(v, dict) and k.startswith("k_"):
            assert k in names, f"{k}: in the PMC summary, not in {pm['tag']}_kernel_stats.csv"


def test_the_committed_profile_is_of_this_code():
    """The profile bench.py will quote was taken with the code in this tree: whoever changes a kernel, the C ABI, the launch loop or the
    evaluator re-runs profiles/run_profile.sh (or this test says so). bench.py itself would fall back to its raw HIP-event figure."""
    from chinesechesszero_amd.build import code_hash
    with open(os.path.join(ROOT, "profiles", "pmc_summary.json")) as f:
        pm = json.load(f)
    assert pm["head"] == code_hash(), (f"profiles/pmc_summary.json was taken with code {pm['head']}, the tree is {code_hash()}: "
                                       "re-run `bash profiles/run_profile.sh rNN trace fetch write sq` (and the c* passes) on a GPU box")
    assert "head_warning" not in pm, pm.get("head_warning")


def test_code_hash_follows_the_kernel_sources(tmp_path, monkeypatch):
    from chinesechesszero_amd import build
    h0 = build.code_hash()
    assert len(h0) == 16 and h0 == build.code_hash()
    # a copy of the package tree with one byte more in one kernel header hashes differently
    import shutil
    dst = tmp_path / "repo" / "chinesechesszero_amd"
    shutil.copytree(os.path.join(ROOT, "chinesechesszero_amd"), dst, ignore=shutil.ignore_patterns("*.so", "__pycache__", ".pytest_cache"))
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "repo" / "include")     # (round 6: the C header and bench.py are hashed too)
    shutil.copy(os.path.join(ROOT, "bench.py"), tmp_path / "repo" / "bench.py")
    monkeypatch.setattr(build, "_HERE", str(dst))
    assert build.code_hash() == h0
    with open(dst / "csrc" / "cczero_kernels.h", "a") as f:
        f.write("\n")
    h1 = build.code_hash()
    assert h1 != h0
    with open(tmp_path / "repo" / "include" / "cczero.h", "a") as f:
        f.write("\n")
    assert build.code_hash() not in (h0, h1)


def test_a_library_older_than_its_sources_is_refused(tmp_path, monkeypatch):
    """The csrc Makefile stamps libcczero.so with a digest of the sources it was built from; `_lib.lib()` refuses a library whose stamp is
    not the digest of the sources in the tree (round 5 ran a whole round of GPU calls through a library built before the last kernel edit)."""
    import shutil
    from chinesechesszero_amd import _lib
    assert _lib.stale_build() is None and _lib.source_hash() == open(_lib.LIB_PATH + ".srchash").read().strip()
    root = tmp_path / "repo"
    shutil.copytree(os.path.join(ROOT, "chinesechesszero_amd"), root / "chinesechesszero_amd", ignore=shutil.ignore_patterns("__pycache__", ".pytest_cache"))
    shutil.copytree(os.path.join(ROOT, "include"), root / "include")
    monkeypatch.setattr(_lib, "_HERE", str(root / "chinesechesszero_amd"))
    monkeypatch.setattr(_lib, "LIB_PATH", str(root / "chinesechesszero_amd" / "libcczero.so"))
    assert _lib.stale_build() is None
    with open(root / "chinesechesszero_amd" / "csrc" / "cczero_kernels.h", "a") as f:
        f.write("// edited after the build\n")
    assert "built from other sources" in _lib.stale_build()
    monkeypatch.setattr(_lib, "_lib", None)
    import pytest
    with pytest.raises(_lib.CczError, match="rebuild"):
        _lib.lib()
    os.remove(root / "chinesechesszero_amd" / "libcczero.so.srchash")
    assert "carries no source stamp" in _lib.stale_build()
