"""One-off soak: tests/test_gpu_fuzz.py's random joint configurations for many more seeds (not part of the suite).
usage: python scripts/fuzz_soak.py first last"""
import sys, time, traceback
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import importlib
import conftest  # noqa: F401  (tests/conftest.py: builds / loads the oracle)
from oracle import oracle as O
O.build(ref=False)
fz = importlib.import_module("test_gpu_fuzz")
fn = fz.test_random_joint_configuration
fn = getattr(fn, "__wrapped__", fn)
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = []
t0 = time.time()
for seed in range(a, b):
    try:
        fn(O, seed)
    except Exception as e:
        bad.append(seed)
        print("seed", seed, "FAILED:", repr(e)[:300]); traceback.print_exc(limit=1)
    if time.time() - t0 > float(sys.argv[3]) if len(sys.argv) > 3 else False:
        print("time budget reached at seed", seed); break
print("done", a, seed + 1, "failures:", bad, "%.1f s" % (time.time() - t0))
