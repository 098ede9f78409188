import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waverange_amd import api
nb = 60
n = 60000 * nb
rs = np.random.RandomState(1)
def plane(kind):
    if kind == "two": return rs.choice(np.array([127, 128], np.uint8), size=n, p=[0.8, 0.2])
    if kind == "one": return np.where(rs.random_sample(n) < 0.9997, 128, rs.randint(120, 136, n)).astype(np.uint8)
    return np.minimum(rs.randint(0, 256, n), rs.randint(64, 320, n)).astype(np.uint8)
for kind in ("two", "one", "noise"):
    ps3 = [plane(kind) for _ in range(3)]
    best = 1e9
    for _ in range(3):
        t = time.time(); api.range_encode_multi(ps3); best = min(best, time.time() - t)
    print("scalar loop of 3, kind %s: %.1f Msym/s" % (kind, 3 * n / best / 1e6))
    for k in (4, 8, 16):
        ps = [plane(kind) for _ in range(k)]
        best = 1e9
        for _ in range(3):
            t = time.time(); api.range_encode_vec(ps); best = min(best, time.time() - t)
        print("vector encoder, %2d planes of kind %s: %.1f Msym/s per thread" % (k, kind, k * n / best / 1e6))
