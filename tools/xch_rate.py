"""Round time of the collective-free combine (hyphy_hip_xch_*, comm.hip) with N real processes on this host — pure host code, no GPU:
python tools/xch_rate.py [N ...]   →   one line per N: microseconds per exchange round (includes ≈ 1 µs of ctypes call)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--rank":
    from hyphy_amd import hip
    rank, world, n, tag = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    x = hip.HostExchange("xch_rate_" + tag, rank, world)
    for i in range(2000):
        x.sum(float(i + rank))
    t0 = time.perf_counter()
    for i in range(n):
        v = x.sum(1.0 + rank)
    t1 = time.perf_counter()
    if rank == 0:
        print(f"{world} ranks: {(t1 - t0) / n * 1e6:.2f} us per exchange round (sum {v}, {n} rounds, {os.cpu_count()} host cores)")
    x.close()
else:
    for w in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
        tag = f"{os.getpid()}_{w}"
        ps = [subprocess.Popen([sys.executable, __file__, "--rank", str(r), str(w), "200000", tag]) for r in range(w)]
        for p in ps:
            p.wait()
