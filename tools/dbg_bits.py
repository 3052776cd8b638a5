import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import safepy_amd as amd
from oracle import safe_oracle as orc
rng = np.random.default_rng(1)
for (n, m, P) in [(8, 2, 10), (70, 3, 10), (300, 70, 40)]:
    a = (rng.uniform(size=(n, n)) < 0.3).astype(np.int64)
    b = (rng.uniform(size=(n, m)) < 0.4).astype(np.float64)
    cn_w, cp_w = orc.run_permutations(a, b, 'sum', P, 3)
    for path in ('gather', 'scatter', 'bits'):
        os.environ['SAFE_HIP_FORCE_PATH'] = path
        cn, cp = amd.run_permutations((a, b, 'sum', P, 3))
        k = amd.Context.default(0).last_kernel()[0]
        print(n, m, P, path, 'neg ok', np.array_equal(cn, cn_w), 'pos ok', np.array_equal(cp, cp_w), k)
