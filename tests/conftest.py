import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def rccl_world1():
    """One 1-rank RCCL process group for the whole session: every data-parallel GPU test shares it.  Creating and
    destroying a communicator per test leaves each destroyed group's internal stream and watchdog to be torn down while
    the next communicator initialises on the same device; one group has no such overlap."""
    import socket
    import torch
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    yield dist.group.WORLD
    torch.cuda.synchronize()
    dist.destroy_process_group()
