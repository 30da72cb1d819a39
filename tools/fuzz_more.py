import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_fuzz as t
from oracle import oracle_py as orc
bad = []
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (40, 400)
for seed in range(lo, hi):
    try:
        t.test_random_case(seed, orc, None)
    except AssertionError as e:
        bad.append((seed, str(e)[:200]))
print("seeds", lo, hi, "failures:", len(bad))
for b in bad[:10]: print(b)
