import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_fuzz as t
from oracle import oracle_py as orc
bad = []
for seed in range(40, 400):
    try:
        t.test_random_case(seed, orc, None)
    except AssertionError as e:
        bad.append((seed, str(e)[:200]))
print("failures:", len(bad))
for b in bad[:10]: print(b)
