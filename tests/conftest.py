import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_libs():
    """Build the CPU oracle (the restatement always; the verbatim reference when /root/reference exists)."""
    from oracle import pyoracle
    pyoracle.build("all")
    return pyoracle


@pytest.fixture(scope="session")
def best_oracle_kind(oracle_libs):
    """'ref' (verbatim-compiled reference) when its binary is present, else the pinned restatement."""
    return "ref" if oracle_libs.available("ref", "array") else "port"


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; GPU tests must run on it and nothing else."""
    # torch first: a shard group over RCCL opens librccl at run time and finds torch's copy if it is loaded; opened before
    # torch (a sub-run that starts with the RCCL test), ROCm's own copy AND torch's end up in one process and abort at exit
    import torch  # noqa: F401
    import fiesta_amd
    lib = fiesta_amd.load()
    if fiesta_amd.device_count() < 1:
        pytest.fail("libfiesta_hip.so loaded but no gfx950 device is usable: GPU tests cannot fall back")
    return lib


@pytest.fixture(params=["rounds", "bulk", "levels"])
def engine(request, monkeypatch):
    """Runs a GPU test once per UpdateESDF engine: frontier rounds only / bulk feature transform wherever the map state
    allows it (and, where it does not, the library's own choice: level engine for small updates, rounds for large ones) /
    level engine for everything its lists hold (maps created without an explicit update_engine take
    fiesta_amd.esdf_map.DEFAULT_UPDATE_ENGINE; the library itself reads no environment).  On fully observed maps all three
    must reproduce the reference exactly."""
    import fiesta_amd.esdf_map as em
    code = {"rounds": 1, "bulk": 2, "levels": 3}[request.param]
    monkeypatch.setattr(em, "DEFAULT_UPDATE_ENGINE", code)
    monkeypatch.setenv("FIESTA_TEST_UPDATE_ENGINE", str(code))  # (spawned workers)
    return request.param
