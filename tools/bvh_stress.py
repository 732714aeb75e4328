import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import ray_tracer_amd as rt
rng = np.random.default_rng(2024)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 400):
    n = int(rng.choice([1, 2, 3, 5, 8, 9, 33, 63, 64, 65, 66, 127, 128, 129, 200, 513, 1000, 3000, 7000]))
    kind = it % 7
    if kind == 0:
        v = rng.uniform(-5, 5, size=(n, 1, 3)) + rng.uniform(-0.5, 0.5, size=(n, 3, 3))
    elif kind == 1:
        v = rng.integers(-3, 4, size=(n, 3, 3)).astype(np.float64)
        v[rng.random(v.shape) < 0.15] = -0.0
    elif kind == 2:
        c = rng.normal(size=(max(1, n // 50), 1, 3)) * 30
        v = c[rng.integers(0, c.shape[0], size=n)] + rng.normal(size=(n, 3, 3)) * 1e-3
    elif kind == 3:
        x = np.sort(rng.exponential(size=n)).cumsum()
        v = np.zeros((n, 3, 3)); v[:, :, 0] = x[:, None] + rng.uniform(0, 1e-3, size=(n, 3)); v[:, 1, 1] = 1; v[:, 2, 2] = 1
    elif kind == 4:
        v = np.repeat(rng.uniform(-1, 1, size=(max(1, n // 7), 3, 3)), 7, axis=0)[:n]
        if v.shape[0] < n: v = np.concatenate([v, rng.uniform(-1, 1, size=(n - v.shape[0], 3, 3))])
    elif kind == 6:      # coordinates so small that centroid differences are denormal or vanish
        v = (rng.integers(-4, 5, size=(n, 3, 3)) * 2.0 ** float(rng.choice([-1074, -1070, -1040, -1022, -600]))).astype(np.float64)
    else:
        v = np.ones((n, 3, 3)); v[:, 1, 1] = 2; v[:, 2, 2] = 2
        v[:, 2, 0] += (rng.integers(0, 3, size=n)) * 2.0 ** -51
    v = np.ascontiguousarray(v.reshape(n, 9))
    h = rt.build_bvh(v, rt.BUILDER_HOST); d = rt.build_bvh(v, rt.BUILDER_DEVICE)
    ok = h["nodes"].tobytes() == d["nodes"].tobytes() and np.array_equal(h["indices"], d["indices"]) and h["depth"] == d["depth"]
    if not ok:
        bad += 1
        print("MISMATCH it", it, "n", n, "kind", kind, flush=True)
print("done, mismatches:", bad)
