"""probe: time the two passes of the minimizer scan under different output configurations (run under rocprofv3 --stats)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import biolib_amd as B
mode = sys.argv[1]
c = B.Context(0)
n = 1_500_000_000
b = c.synth(42, n, 150)
cap = 200_000_000
v = c.empty_u64(cap) if mode in ("all", "vp", "v") else None
p = c.empty_u64(cap) if mode in ("all", "vp") else None
h = c.empty_u64(cap) if mode in ("all",) else None
for _ in range(4):
    r = b.minimizers_raw(31, 11, 42, B.FLAG_CANONICAL | B.FLAG_SYNC, values=v, positions=p, hashes=h, capacity=cap if mode != "none" else 0)
print(mode, r.count)
