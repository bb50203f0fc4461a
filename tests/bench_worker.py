"""Child process of tests/test_bench_dp_gpu.py: waits (without touching the GPU) for a go-file, then runs bench.py in-process
with the remaining arguments -- keeps the number of processes on the card low while the data-parallel children are running.

    python tests/bench_worker.py <go-file> <bench.py arguments...>"""
import os
import runpy
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go = sys.argv[1]
t0 = time.time()
while not os.path.exists(go):
    if time.time() - t0 > 780:
        raise SystemExit("bench_worker: no go-file after 780 s")
    time.sleep(0.2)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
