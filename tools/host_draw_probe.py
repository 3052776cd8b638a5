"""Diagnostic: host draw-stream speed inside a Python process, before and after a HIP context
exists and while the GPU is busy (is the draw thread slowed down by the runtime?)."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safepy_amd import backend as be

vals = np.arange(3789, dtype=np.int64)
def probe(tag):
    ts = []
    for _ in range(5):
        t = time.perf_counter(); be.rng_permutations_host(0, vals, 128); ts.append(1e3 * (time.perf_counter() - t))
    print('%-28s 128 x 3789 draws+swaps: min %.2f ms  median %.2f ms' % (tag, min(ts), sorted(ts)[2]))

probe('no HIP context')
ctx = be.Context.default(0)
probe('HIP context created')
import safepy_amd
from safepy_amd import workloads
data = workloads.costanzo_surrogate(seed=0, m=2048)
sf = safepy_amd.SAFE(verbose=False)
sf.graph = safepy_amd.LayoutGraph(data['xy'], data['edge_u'], data['edge_v'], length=data['length'])
sf.define_neighborhoods()
probe('after some GPU work')
# GPU busy in another thread
b = data['attributes']
def gpu_work():
    sf2 = sf
    sf2.load_attributes(attribute_file=b)
    sf2.random_seed = 0
    for _ in range(3):
        sf2.compute_pvalues(how='randomization', num_permutations=1000, verbose=False)
th = threading.Thread(target=gpu_work); th.start()
time.sleep(0.05)
probe('while compute_pvalues runs')
th.join()
