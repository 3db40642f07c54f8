"""Pairs/s of the two-lane step (train.PairLanes) against GPU_MAX_HW_QUEUES: the HIP runtime multiplexes its streams
onto that many hardware queues, and streams that share a queue run back to back.  One process per setting (the variable
is read when the runtime starts).
    python profiles/hw_queue_sweep.py [4 8 16 32 64]"""
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
for q in (sys.argv[1:] or ["4", "8", "16", "32", "64"]):
    env = dict(os.environ, GPU_MAX_HW_QUEUES=q)
    r = subprocess.run([sys.executable, os.path.join(here, "lanes_host_trace.py"), "2", "20"], env=env,
                       capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if "pairs/s" in l]
    print("GPU_MAX_HW_QUEUES=%-3s %s" % (q, line[0] if line else "FAILED: " + r.stderr[-300:]))
    sys.stdout.flush()
