import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the tests use tiny nets of many shapes: skip MIOpen's per-shape find step (bench.py keeps it on)
os.environ.setdefault("CCZ_MIOPEN_FIND", "0")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _ensure_built():
    """A fresh checkout has no libcczero.so (built artefacts are git-ignored): build it once if hipcc is here."""
    so = os.path.join(ROOT, "chinesechesszero_amd", "libcczero.so")
    stale = False
    if os.path.exists(so):
        from chinesechesszero_amd import _lib
        stale = _lib.stale_build() is not None      # built before the last kernel edit (libcczero.so.srchash vs the sources in the tree)
    if (stale or not os.path.exists(so)) and os.path.exists("/opt/rocm/bin/hipcc"):
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()


def pytest_addoption(parser):
    parser.addoption("--rules-probe", default=None, metavar="DIR",
                     help="directory written by tools/probe_cchess.py where a real `cchess` is installed (preset.json + cchess_golden.npz): "
                          "tests/test_gpu_rules_probe.py then answers 'does the engine match MY cchess' on the GPU "
                          "(default: the probe is run against the CPU oracle posing as cchess)")


@pytest.fixture(scope="session")
def rules_probe_dir(request):
    d = request.config.getoption("--rules-probe")
    if d is not None and not (os.path.exists(os.path.join(d, "preset.json")) and os.path.exists(os.path.join(d, "cchess_golden.npz"))):
        raise pytest.UsageError(f"--rules-probe {d}: preset.json / cchess_golden.npz not found (run tools/probe_cchess.py --out {d})")
    return d


def pytest_configure(config):
    _ensure_built()
    if os.environ.get("CCZ_LIB"):  # run the suite against a diagnostic build (build/diag/libcczero_bounds.so: `make bounds`)
        from chinesechesszero_amd import _lib
        _lib.LIB_PATH = os.path.join(ROOT, "build", "diag", os.environ["CCZ_LIB"])
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU check")


@pytest.fixture(scope="session")
def golden():
    import json

    import numpy as np

    g = os.path.join(ROOT, "tests", "golden")
    data = dict(np.load(os.path.join(g, "reference_search.npz")))
    with open(os.path.join(g, "reference_search.json")) as f:
        meta = json.load(f)
    with open(os.path.join(g, "action_table.txt")) as f:
        table = f.read().split()
    return {"data": data, "meta": meta, "table": table}


@pytest.fixture
def rules_of_case():
    """Install the `legal_moves` order a golden case was generated with (oracle and, if asked, the product's host mirror),
    and restore the defaults afterwards. Returns the rank permutation (or None)."""
    import numpy as np

    import oracle
    installed = []

    def install(case, product: bool = False, both: bool = False):
        from golden_cases import case_order
        L = oracle.lib()
        rank, trank = case_order(case, [L.xq_move_from(i) for i in range(2086)], [L.xq_move_to(i) for i in range(2086)])
        oracle.set_rules(move_rank=rank, type_rank=trank)
        installed.append("oracle")
        if product:
            from chinesechesszero_amd import tools
            tools.set_rules(move_rank=rank, type_rank=trank)
            installed.append("product")
        return (rank, trank) if both else rank

    yield install
    oracle.set_rules()
    if "product" in installed:
        from chinesechesszero_amd import tools
        tools.set_rules()
