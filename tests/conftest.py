import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Fixtures are plain arrays written by oracle/make_golden.py; allow_pickle stays False."""
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden


# ---- GEMM arithmetic: the GPU suites run under all three reference-precision arithmetics -----------------------------------
# "h3" (library default): the composite stacks multiply two fp16 pieces per fp32 operand under tracked power-of-two scales (three
# f16 MFMAs, fp32 accumulation; every other GEMM as b6); "b6": three bf16 pieces per fp32 operand, six bf16 MFMAs (csrc/ctn_gemm_b3.h);
# "fp32": fp32-MFMA kernels (bit-exact fp32 FMA chains).  Every limit asserted in the GPU tests is the fp32 limit and applies
# UNCHANGED to all three (TOL_SCALE = 1).  tests/test_gpu_h3.py tests the h3 entry points themselves.
ARITH = {"name": "fp32"}
TOL_SCALE = {"fp32": 1.0, "b6": 1.0, "h3": 1.0}
DEFAULT_ARITH = "h3"


def tol_scale():
    return TOL_SCALE[ARITH["name"]]


def set_arith(name):
    import conv_tasnet_amd as ctn
    ctn.set_gemm_arith(name)
    ARITH["name"] = name


@pytest.fixture(params=["h3", "b6", "fp32"])
def gemm_arith(request):
    set_arith(request.param)
    yield request.param
    set_arith(DEFAULT_ARITH)          # the library default


@pytest.fixture
def fp32_only(gemm_arith):
    """For tests of the fp32-MFMA kernel families and plans themselves."""
    if gemm_arith != "fp32":
        pytest.skip("exercises the fp32-MFMA kernels")


# ---- data-parallel GPU test: children are started before anything in this process touches the GPU -------------------
_DP = {}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def pytest_collection_finish(session):
    """Spawn tests/dp_worker.py (2 ranks + the single-process reference) if the DP GPU test was selected and a GPU exists.
    torch.cuda.device_count() does not initialise the GPU; the children run beside the other tests and are collected by
    tests/test_dp_gpu.py."""
    _spawn_bench_children(session)
    if not any(it.nodeid.endswith("test_two_rank_product_training_matches_single_process") for it in session.items):
        return
    import subprocess
    import tempfile
    import torch
    if torch.cuda.device_count() < 1:
        return
    tmp = tempfile.mkdtemp(prefix="ctn_dp_")
    worker = os.path.join(ROOT, "tests", "dp_worker.py")
    procs, outs, logs = {}, {}, {}
    port2, port1 = _free_port(), _free_port()
    for name, rank, world, port in (("rank0", 0, 2, port2), ("rank1", 1, 2, port2), ("single", 0, 1, port1)):
        outs[name] = os.path.join(tmp, name + ".pt")
        logs[name] = os.path.join(tmp, name + ".log")
        procs[name] = subprocess.Popen([sys.executable, worker, str(rank), str(world), port, outs[name]],
                                       stdout=open(logs[name], "w"), stderr=subprocess.STDOUT, cwd=ROOT)
    _DP["children"] = (procs, outs, logs)


def _spawn_bench_children(session):
    """`bench.py --gpus 2 --steps 3 --warmup 1 --config tiny` as two env-rendezvous ranks on cuda:0 over gloo."""
    if not any(it.nodeid.endswith("test_bench_two_ranks_prints_one_complete_line") for it in session.items):
        return
    import subprocess
    import tempfile
    import torch
    if torch.cuda.device_count() < 1:
        return
    tmp = tempfile.mkdtemp(prefix="ctn_bench_dp_")
    port = _free_port()
    procs, outs, logs = {}, {}, {}
    for rank in (0, 1):
        name = "rank%d" % rank
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   CTN_DIST_BACKEND="gloo")
        outs[name], logs[name] = os.path.join(tmp, name + ".out"), os.path.join(tmp, name + ".log")
        procs[name] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "bench_worker.py"), os.path.join(tmp, "go"), "--gpus", "2",
                                        "--steps", "3", "--warmup", "1", "--config", "tiny"],
                                       stdout=open(outs[name], "w"), stderr=open(logs[name], "w"), cwd=ROOT, env=env)
    _DP["bench"] = (procs, outs, logs)
    _DP["bench_go"] = os.path.join(tmp, "go")


@pytest.fixture
def bench_children():
    if "bench" not in _DP:
        pytest.skip("no GPU visible at collection time: the bench ranks were not started")
    if "children" in _DP:                       # at most 6 processes may use the card: let the data-parallel children finish first
        for p in _DP["children"][0].values():
            p.wait(timeout=800)
    open(_DP["bench_go"], "w").close()          # the two ranks wait for this file before they import torch
    return _DP["bench"]


@pytest.fixture
def dp_children():
    if "children" not in _DP:
        pytest.skip("no GPU visible at collection time: the data-parallel children were not started")
    return _DP["children"]
