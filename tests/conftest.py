import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def torch_hip_first():
    """When torch is going to share the process (device buffers for the all-gather), let it bring up the HIP
    runtime BEFORE libycge_hip.so is loaded, so that both resolve to one libamdhip64 (bench.py does the same)."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except Exception:
        pass


@pytest.fixture(scope="session")
def product_lib(torch_hip_first):
    """libycge_hip.so — must exist (built by __graft_entry__.build()); never substituted."""
    from yetanotherconsolegameengine_amd import abi, build
    build.build_library()
    return abi.load_library()


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    return oracle_binding
