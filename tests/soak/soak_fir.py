"""One-off soak: tests/test_gpu_parity.py::test_fir_shapes_fuzz with shifted seeds (other shapes, other data)."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import test_gpu_parity as T
orig = np.random.default_rng
t_end = time.time() + float(sys.argv[1] if len(sys.argv) > 1 else 60)
k = 0
while time.time() < t_end:
    k += 1
    np.random.default_rng = lambda s=None, _k=k: orig(None if s is None else int(s) + 7919 * _k)
    T.test_fir_shapes_fuzz()
    print("round", k, "ok", flush=True)
print("soak_fir: %d rounds x 80 shapes, all bit-identical" % k)
