"""GPU: `bench.py --gpus 2` as the driver launches it -- two ranks under torch.distributed env rendezvous -- rehearsed on ONE GPU
with the gloo backend (RCCL needs distinct devices; the ranks share cuda:0), on BASELINE configs[0] (tiny) so that the whole
line, `roofline` and `cpu_baseline` included, takes seconds.  The two ranks are child processes started by conftest.py BEFORE
this process touches the GPU (tests/conftest.py: pytest_collection_finish) and only collected here."""
import json

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_bench_two_ranks_prints_one_complete_line(bench_children):
    procs, outs, logs = bench_children
    for name, p in procs.items():
        rc = p.wait(timeout=800)
        assert rc == 0, "%s exited with %d:\n%s" % (name, rc, open(logs[name]).read()[-3000:])
    lines0 = [l for l in open(outs["rank0"]).read().splitlines() if l.strip().startswith("{")]
    lines1 = [l for l in open(outs["rank1"]).read().splitlines() if l.strip().startswith("{")]
    assert len(lines0) == 1 and lines1 == [], "rank 0 prints ONE JSON line, the other ranks none"
    j = json.loads(lines0[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["unit"] == "utterances/sec" and j["value"] > 0 and j["dtype"].startswith("f32 (") and j["vs_baseline"] is None
    d = j["distributed"]                                   # what the collective library saw (RCCL on the driver's node; gloo here)
    assert d["rccl_ranks"] == 2 and len(d["ranks"]) == 2 and d["backend"] == "gloo" and d["gradient_allreduce_bytes_per_step"] > 0
    assert j["roofline"]["gradient_allreduce"]["exposed_us_per_step"] >= 0
    assert j["config"]["global_batch"] == 4 and j["config"]["parallelism"] == "dp2"
    assert abs(j["value"] - 4 * 3 / (j["ms_per_step"] * 3e-3)) < 0.01 * j["value"]      # whole-job utterances over the timed region
    assert j["mean_loss"] == j["mean_loss"] and abs(j["mean_loss"]) < 1e3                # finite
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["families"] and r["step"]["ms_per_step"] == j["ms_per_step"]
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["timed_steps"] >= 5
