#!/usr/bin/env python3
"""bench.py -- node-attribute enrichments/s of compute_pvalues (permutation test) on MI355X.

A "step" is one full pass of the hot path over one batch of synthetic input that is
already resident in HBM: whole-matrix statistics for the dispatch rule, the seeded legacy
MT19937 permutation stream (host) and its upload, the permutation-test kernel with the
fused p-value / NES / binarisation epilogue, and -- for N > 1 -- the RCCL all-gather of the
NES matrix.  Workload = BASELINE.json configs[1]: Costanzo-2016-shaped network (3971
nodes, default metric) x 4373 GO-BP-like binary attributes x 1000 permutations, on a
seeded surrogate (safe-data is not available offline; safepy_amd/workloads.py).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1: one rank per GPU (launched by torch.distributed.run; when WORLD_SIZE is not set the script
starts that launcher itself as a child process, before anything touches a GPU, and relays rank
0's line and the exit code); every rank owns its own 4373-attribute shard (weak scaling: per-GPU
work fixed), the network and the permutation stream are replicated, results are all-gathered
over RCCL.  The step is sharding.randomization_step -- the function the product's sharded driver
(safepy_amd.run_batch) runs.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # (200 steps: 0.7 s of timed work -- a pre-empted host thread costs one step ~5 ms about once in 500 steps on the shared bench
    # hosts, tools/probe/trace_outlier.py; with 20 steps one such step moved the mean by 7 %)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=50)        # the first ~50 steps of a fresh process run 0.1 ms slower (step_ms_mean_by_quarter_of_run)
    ap.add_argument('--perms', type=int, default=1000)
    ap.add_argument('--nodes', type=int, default=3971)
    ap.add_argument('--attrs', type=int, default=4373)
    ap.add_argument('--metric', default='shortpath_weighted_layout')
    ap.add_argument('--radius', type=float, default=0.1)
    ap.add_argument('--cpu-perms', type=int, default=40, help='permutations timed for the CPU baseline (0 = skip)')
    ap.add_argument('--extras', type=int, default=1, help='also time the HBM-bound kernels (K1 distance, K4 hypergeometric)')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak',
                    help='N > 1: weak = every rank its own block of --attrs columns (the default line); strong = --attrs columns '
                         'split over the ranks (np.array_split, safe.py:1339): `--scaling strong --perms 10000` is BASELINE configs[2]')
    ap.add_argument('--workload', choices=['cfg1', 'cfg4'], default='cfg1',
                    help='cfg1 = the Costanzo-shaped binary workload of configs[1]/[2]; cfg4 = BASELINE configs[4]: 20 000 nodes, '
                         '6250 quantitative f64 columns per rank x 1000 permutations (matrix-core kernel)')
    ap.add_argument('--multi-extras', type=int, default=1,
                    help='N > 1: after the timed weak-scaling line also time configs[2] (strong scaling, 10 000 permutations) and one '
                         'step of the configs[4] rank share, reported inside the same JSON line')
    return ap.parse_args()


def _by_decile(step_ms, values):
    """Mean of `values` over the steps of each decile of the step-time distribution (None where a value is missing)."""
    import numpy as np
    pairs = sorted((s, v) for s, v in zip(step_ms, values) if v is not None)
    if len(pairs) < 10:
        return None
    return [round(float(np.mean([v for _, v in chunk])), 3) for chunk in np.array_split(np.asarray(pairs, dtype=object), 10)]


def _cpu_model():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def effective_cores():
    from safepy_amd import backend
    return backend.effective_cores()


def cpu_baseline(a_dense, b, sample_perms):
    """The oracle's permutation loop (same NumPy calls as the reference: np.where x2,
    int64 x float np.dot, fancy-index row permutation, <= / >= accumulation) on the host,
    BLAS threads = all cores, bounded to `sample_perms` permutations after one warm-up."""
    import numpy as np
    from oracle import safe_oracle as orc
    n, m = b.shape
    cores = effective_cores()
    blas = None
    try:                                           # BLAS threads = the cores we may really use (oversubscribing a CPU quota gets the process throttled)
        from threadpoolctl import threadpool_limits, threadpool_info
        limiter = threadpool_limits(limits=cores)
    except Exception:
        limiter = None
    t0 = time.perf_counter()
    orc.run_permutations(a_dense, b, 'sum', 1, 0)              # warm-up (first np.dot is ~2.5x slower)
    t_warm = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.run_permutations(a_dense, b, 'sum', sample_perms, 0)   # computes the observed score once + P permutations
    dt = time.perf_counter() - t0
    value = n * m * sample_perms / dt
    if limiter is not None:
        blas = [(i.get('internal_api'), i.get('num_threads')) for i in threadpool_info()]
        limiter.restore_original_limits()
    return {'value': value, 'unit': 'enrichments/s', 'cores': cores, 'host_cpus': os.cpu_count(), 'kind': 'port',
            'sample': '%d permutations of the same %dx%d workload after 1 warm-up (%.1f s; warm-up %.1f s), NumPy/SciPy oracle'
                      % (sample_perms, n, m, dt, t_warm),
            'seconds_per_permutation': dt / sample_perms, 'blas': blas}


def kernel_source_sha256():
    """Hash of the kernel sources the library is built from (safepy_amd/csrc/*.hip|*.h|*.cpp + include/):
    what the committed PMC figures are stamped with (tools/update_profiles.py, tools/pmc_bits.sh)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, 'safepy_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(ROOT, 'safepy_amd', 'csrc', '*.h')) +
                   glob.glob(os.path.join(ROOT, 'safepy_amd', 'csrc', '*.cpp')) + glob.glob(os.path.join(ROOT, 'include', '*.h')))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()


def _stamp_is_stale(stamp, kernel_name):
    """A committed PMC figure is stale when the kernel sources changed after it was profiled, or it belongs to another kernel."""
    if not isinstance(stamp, dict):
        return True
    if kernel_name is not None and not str(stamp.get('kernel', '')).startswith(kernel_name):
        return True
    return stamp.get('kernel_source_sha256') != kernel_source_sha256()


def pmc_traffic(kernel_name, with_stamp=False):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate --pmc
    passes of this same bench command; tools/rocpd_counters.py).  PMC counters cannot be
    collected from inside the timed process, so this is the offline measurement; None if the
    profile has no entry for the kernel.  with_stamp: (bytes, {commit, stale}) -- stale = the kernel
    sources differ from the ones that were profiled."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
            doc = json.load(f)
        entry = doc.get(kernel_name)
        val = None if entry is None else entry['hbm_bytes_per_launch']
        stamp = doc.get('_stamp')
    except (OSError, ValueError, KeyError):
        val, stamp = None, None
    if not with_stamp:
        return val
    stamp = stamp if isinstance(stamp, dict) else {}
    return val, {'profiled_at_commit': stamp.get('commit'), 'stale': _stamp_is_stale(dict(stamp, kernel=kernel_name), kernel_name)}


def mfma_pipe_busy(kernel_name):
    """Fraction of the matrix-core pipes' cycles the kernel kept busy, from the committed PMC pass (profiles/pmc_mfma.json:
    SQ_VALU_MFMA_BUSY_CYCLES over GRBM_GUI_ACTIVE x SIMDs; tools/pmc_mfma.sh + tools/update_profiles.py), with its stamp; None
    when there is no such file."""
    path = os.path.join(ROOT, 'profiles', 'pmc_mfma.json')
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    return {'sum': d.get('mfma_pipe_busy_sum'), 'z_score': d.get('mfma_pipe_busy_zscore'), 'stale': _stamp_is_stale(d, None),
            'commit': d.get('commit'), 'source': d.get('_source')}


VALU_PEAK_WAVE_INSTR_PER_SIMD_US = 830.0     # measured: tools/ubench/valu_banks.hip (v_bitop3_b32, two and four waves per SIMD, any operand banks)


def binding_resources(num_cu, kernel_name=None, busy_ms=None):
    """Utilisation of the resources that actually bound the dominant kernel (it is neither HBM- nor MFMA-bound), from the
    committed PMC passes of this bench command (profiles/pmc_bits.json; counters cannot be read inside the timed process):
    LDS pipe busy cycles per CU and VALU issue cycles per SIMD over the kernel's GPU cycles.  Stamped with the commit and
    kernel they were profiled at; "stale": true when the kernel sources changed since (or the dominant kernel is another one)."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_bits.json')) as f:
            c = json.load(f)
        gpu_cycles = c['GRBM_GUI_ACTIVE'] / 8.0               # the counter is summed over the 8 XCDs
        wall = {}
        if busy_ms:
            # the launches of a step overlap on two streams, so fractions over the SUM of their durations (below) understate how
            # busy the chip is while the kernel runs: the same instruction counts over the union of the launch intervals of THIS run
            steps = float(c.get('steps_profiled', 1))
            rate = c['SQ_INSTS_VALU'] / steps / (4 * num_cu) / (busy_ms * 1e3)
            wall = {'valu_wave_instr_per_step': c['SQ_INSTS_VALU'] / steps,
                    'valu_wave_instr_per_simd_and_us_while_busy': rate,
                    'valu_issue_frac_of_measured_peak_while_busy': rate / VALU_PEAK_WAVE_INSTR_PER_SIMD_US,
                    'lds_wave_instr_per_cu_and_us_while_busy': c['SQ_INSTS_LDS'] / steps / num_cu / (busy_ms * 1e3),
                    'note': 'diagnostic builds (DESIGN.md 4, K5a): without any LDS gather the kernel takes the same time; the bound is the issue '
                            'rate of four in-order waves per SIMD (a lone wave issues one instruction per ~6 clocks)'}
        return {'kernel': c['kernel'], **wall,
                'lds_pipe_busy_frac': c['SQ_LDS_IDX_ACTIVE'] / num_cu / gpu_cycles,
                'lds_bank_conflict_share_of_busy': c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'],
                'valu_issue_frac': c['SQ_INSTS_VALU'] / (4 * num_cu) / c['valu_wave_insts_per_clock_per_simd_sustained'] / gpu_cycles,
                'waves_issuing_parked_stalled': [c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES'], c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'],
                                                 c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']],
                'source': c['_source'], 'profiled_at_commit': c.get('commit'),
                'stale': _stamp_is_stale(c, kernel_name)}
    except (OSError, ValueError, KeyError, ZeroDivisionError):
        return None


class ClockSampler:
    """Engine / memory clock of THIS GPU while a kernel loop runs, from the amdgpu driver's sysfs files (pp_dpm_sclk / pp_dpm_mclk: the
    level marked '*'), sampled every 2 ms on a thread of its own.  The streaming kernels' rates differ by +-10 % from box to box:
    the clocks they ran at belong beside them.  (No rocm-smi child process: a process that has initialised the GPU must not exec.)
    The card is found by its PCI address; without it (or without sysfs) the report is None."""

    def __init__(self, torch):
        import glob
        self.dev = None
        try:
            p = torch.cuda.get_device_properties(torch.cuda.current_device())
            want = '%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
            for path in glob.glob('/sys/class/drm/card*/device'):
                if os.path.basename(os.path.realpath(path)) == want and os.path.exists(os.path.join(path, 'pp_dpm_sclk')):
                    self.dev = path
        except Exception:
            self.dev = None
        self.samples = {'sclk': [], 'mclk': []}
        self._stop = False
        self._thread = None

    def _read(self, key):
        try:
            with open(os.path.join(self.dev, 'pp_dpm_' + key)) as f:
                for ln in f:
                    if ln.strip().endswith('*'):
                        return int(''.join(ch for ch in ln.split(':', 1)[1] if ch.isdigit()))
        except (OSError, ValueError, IndexError):
            pass
        return None

    def __enter__(self):
        if self.dev is not None:
            import threading

            def loop():
                while not self._stop:
                    for key in self.samples:
                        v = self._read(key)
                        if v is not None:
                            self.samples[key].append(v)
                    time.sleep(0.002)
            self._thread = threading.Thread(target=loop, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._thread is not None:
            self._thread.join()

    def report(self):
        if self.dev is None or not self.samples['sclk']:
            return None
        s = self.samples
        return {'sclk_mhz_max': max(s['sclk']), 'sclk_mhz_median': sorted(s['sclk'])[len(s['sclk']) // 2],
                'mclk_mhz_max': max(s['mclk']) if s['mclk'] else None, 'samples': len(s['sclk'])}


def hbm_kernels(ctx, torch, np, be):
    """The two HBM-bound kernels of the path at BASELINE.json configs[3] size, timed with HIP
    events on the context stream: K1 fused all-pairs distance + threshold writing the reference's
    int64 [N,N] layout (N = 20 000: 3.2 GB), and K4 hypergeometric tail + NES + binarisation
    (N = 20 000 x M = 10 000 binary attributes = configs[3]: matrix-core counts leave packed u16
    counts, the table of the distinct (n, K) pairs follows, and k_hyp_emit -- the kernel timed here --
    streams p / nes / nes_binary out; the whole call is timed beside it)."""
    out = {}
    n = 20000
    rng = np.random.default_rng(4)
    xy = rng.uniform(size=(n, 2))
    t_xy = torch.from_numpy(xy).to('cuda')
    t_mask = torch.empty((n, n), dtype=torch.int64, device='cuda')
    nr = 0.1 * (xy[:, 0].max() - xy[:, 0].min())
    for _ in range(2):
        ctx.euclidean_dense(t_xy.data_ptr(), n, nr, t_mask.data_ptr(), None)
    with ClockSampler(torch) as clk:                  # (50 launches while sampling, the first 5 timed as before)
        ctx.timer_start()
        reps = 5
        for _ in range(reps):
            ctx.euclidean_dense(t_xy.data_ptr(), n, nr, t_mask.data_ptr(), None)
        ms = ctx.timer_stop_ms() / reps
        for _ in range(45):
            ctx.euclidean_dense(t_xy.data_ptr(), n, nr, t_mask.data_ptr(), None)
        ctx.sync()
    alg = 16 * n + 8 * n * n
    out['k_euclid_dense'] = {'bound': 'hbm', 'workload': 'N=%d, int64 [N,N] membership (reference layout)' % n,
                             'kernel_ms': ms, 'algorithmic_bytes': alg, 'achieved': alg / ms / 1e6, 'peak': HBM_PEAK_GBS,
                             'unit': 'GB/s', 'frac': alg / ms / 1e6 / HBM_PEAK_GBS, 'traffic': pmc_traffic('k_euclid_dense'),
                             'clocks_while_running': clk.report()}
    del t_mask
    m = 10000                                # configs[3]: 20 000 nodes x 10 000 binary attributes
    b = (rng.uniform(size=(n, m)) < 0.01).astype(np.float32)
    nbr = be.Neighborhoods.euclidean(ctx, xy, nr)
    attr = be.Attributes.from_host(ctx, b)
    bufs = [torch.empty((n, m), dtype=torch.float64, device='cuda') for _ in range(3)] + \
           [torch.empty((m,), dtype=torch.float64, device='cuda')]
    ptrs = [t.data_ptr() for t in bufs]
    be.hypergeom(ctx, nbr, attr, 0.05, ptrs)
    be.hypergeom(ctx, nbr, attr, 0.05, ptrs)
    ctx.sync()
    with ClockSampler(torch) as clk:
        t0 = time.perf_counter()
        be.hypergeom(ctx, nbr, attr, 0.05, ptrs)
        ctx.sync()
        call_ms = 1e3 * (time.perf_counter() - t0)
        name, ms, _ = ctx.last_kernel()
        for _ in range(10):
            be.hypergeom(ctx, nbr, attr, 0.05, ptrs)
        ctx.sync()
    if name == 'k_hyp_emit':                   # split form: streams p, nes, nes_binary out; reads the packed u16 counts (12 B per 6 elements per row position)
        n_pos = 256 * ((n + 255) // 256)
        alg = n * m * 8 * 3 + n_pos * ((m + 191) // 192) * 384
    elif name.startswith('k_permtest_mfma'):     # matrix-core counts + table lookup: write p, nes, nes_binary; read the 0/1 planes and the membership blocks
        alg = n * m * 8 * 3 + (n + 1) * m + be.block_count(nbr) * 1024
    elif name.startswith('k_counts_bits'):     # fused: counts never reach memory; write p, nes, nes_binary; read bit words + member ids
        alg = n * m * 8 * 3 + 8 * (n + 1) * ((m + 63) // 64) + 4 * int(nbr.nnz)
    else:
        alg = n * m * 8 * 4                   # read X, write p, nes, nes_binary
    out[name] = {'bound': 'hbm', 'workload': 'N=%d x M=%d binary attributes, %d members per neighborhood on average'
                                             % (n, m, int(nbr.nnz / n)),
                 'kernel_ms': ms, 'algorithmic_bytes': alg, 'achieved': alg / ms / 1e6, 'peak': HBM_PEAK_GBS,
                 'unit': 'GB/s', 'frac': alg / ms / 1e6 / HBM_PEAK_GBS, 'traffic': pmc_traffic(name), 'compute_pvalues_call_ms': call_ms,
                 'enrichments_per_s_call': n * m / (call_ms * 1e-3), 'clocks_while_running': clk.report()}
    attr.close()
    nbr.close()
    return out


def dropin_extras(np, torch):
    """SAFE.compute_pvalues() -- the call BASELINE.json's metric names (safe.py:432) -- as a notebook makes it: NumPy matrix on
    the instance in, results on the instance out, at configs[1] (permutation test, seeded) and configs[3] (hypergeometric).
    (a) the call alone: results stay on the device until they are read (lazy_outputs, the default); (b) the call + reading
    `nes` and `nes_binary` as host float64 arrays.  The repeated call reuses the context's buffers and the membership's derived
    structures; `first_call_after_define_ms` is the first compute_pvalues after a new define_neighborhoods (it also builds them)."""
    import safepy_amd
    from safepy_amd import backend as be, workloads
    out = {}

    def measure(tag, graph, metric, b, kw, reps):
        sf = safepy_amd.SAFE(verbose=False)
        sf.random_seed = 0
        sf.graph = graph
        sf.define_neighborhoods(node_distance_metric=metric, neighborhood_radius=0.1)
        sf.node2attribute = b
        import logging
        logging.disable(logging.WARNING)
        t0 = time.perf_counter()
        sf.compute_pvalues(**kw)
        first = 1e3 * (time.perf_counter() - t0)
        sf.compute_pvalues(**kw)
        call = []
        for _ in range(reps):                                   # (a) the call alone: the previous results were never read, their
            t0 = time.perf_counter()                            #     device buffers go back to the context's pool
            sf.compute_pvalues(**kw)
            call.append(1e3 * (time.perf_counter() - t0))
        both, read, release = [], [], []
        for _ in range(reps):                                   # (b) the call + nes and nes_binary as host arrays
            t0 = time.perf_counter()
            sf.nes = None                                       # what a repeated call also pays: the previous call's host arrays
            sf.nes_binary = None                                # (2 x 8NM bytes) go back to the OS
            t1 = time.perf_counter()
            sf.compute_pvalues(**kw)
            t2 = time.perf_counter()
            nes, nb = np.asarray(sf.nes), np.asarray(sf.nes_binary)
            t3 = time.perf_counter()
            release.append(1e3 * (t1 - t0))
            both.append(1e3 * (t3 - t1))
            read.append(1e3 * (t3 - t2))
            del nes, nb
        ctx = be.Context.default(0)
        t0 = time.perf_counter()
        attr = be.Attributes.from_host(ctx, b)
        ctx.sync()
        up = 1e3 * (time.perf_counter() - t0)
        attr.close()
        logging.disable(logging.NOTSET)
        n, m = b.shape
        out[tag] = {'compute_pvalues_ms': float(np.median(call)), 'compute_pvalues_ms_min_max': [float(min(call)), float(max(call))],
                    'read_nes_and_nes_binary_ms': float(np.median(read)), 'call_plus_read_ms': float(np.median(both)),
                    'release_previous_host_results_ms': float(np.median(release)),
                    'first_call_after_define_ms': first, 'calls_timed': reps,
                    'phases': {'upload_attributes_ms': up, 'upload_bytes': int(b.nbytes), 'download_bytes': int(2 * 8 * n * m),
                               'kernel_ms_last_call': float(ctx.last_kernel()[1] * max(int(ctx.last_kernel()[2]), 1)), 'kernel': ctx.last_kernel()[0]},
                    'shape': [int(n), int(m)], 'kwargs': kw}
        del sf

    def notebook_flow(tag, graph, metric, b, kw):
        """The reference's usage pattern (examples/Example_3_Scatterplot_annotation.ipynb:73,104,153): a NEW instance,
        define_neighborhoods -> load_attributes -> compute_pvalues (its first call) -> read `nes` -- every step timed once, as a
        notebook pays them.  (The process is warm: HIP runtime, code objects and the context exist -- the first-ever call of a
        process additionally loads them, ~0.25 s, before any of this.)"""
        import logging
        logging.disable(logging.WARNING)
        runs = []
        for _ in range(3):
            sf = safepy_amd.SAFE(verbose=False)
            sf.random_seed = 0
            sf.graph = graph
            t0 = time.perf_counter()
            sf.define_neighborhoods(node_distance_metric=metric, neighborhood_radius=0.1)
            t1 = time.perf_counter()
            sf.load_attributes(attribute_file=b)
            t2 = time.perf_counter()
            sf.compute_pvalues(**kw)
            t3 = time.perf_counter()
            nes = np.asarray(sf.nes)
            t4 = time.perf_counter()
            runs.append([1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), 1e3 * (t4 - t0)])
            del sf, nes
        logging.disable(logging.NOTSET)
        med = [float(x) for x in np.median(np.asarray(runs), axis=0)]
        out[tag + '_notebook_flow'] = {'define_neighborhoods_ms': med[0], 'load_attributes_ms': med[1], 'first_compute_pvalues_ms': med[2],
                                       'read_nes_ms': med[3], 'total_ms': med[4], 'first_instance_total_ms': runs[0][4], 'instances': len(runs)}

    data = workloads.costanzo_surrogate(seed=0)
    notebook_flow('configs1', safepy_amd.LayoutGraph(data['xy'], data['edge_u'], data['edge_v'], length=data['length']),
                  'shortpath_weighted_layout', data['attributes'], dict(how='randomization', num_permutations=1000))
    measure('configs1_randomization', safepy_amd.LayoutGraph(data['xy'], data['edge_u'], data['edge_v'], length=data['length']),
            'shortpath_weighted_layout', data['attributes'], dict(how='randomization', num_permutations=1000), 10)
    del data
    n, m = 20000, 10000
    b = (np.random.default_rng(5).uniform(size=(n, m)) < 0.01).astype(np.float32)
    notebook_flow('configs3', safepy_amd.LayoutGraph(workloads.uniform_layout(4, n)), 'euclidean', b, {})
    measure('configs3_hypergeometric', safepy_amd.LayoutGraph(workloads.uniform_layout(4, n)), 'euclidean', b, {}, 3)
    # the same 0/1 matrix handed over as uint8 (additive: SAFE_DTYPE_U8): a quarter of the bytes over the link, identical results
    # (tests/test_gpu_u8.py); the f32 figure above stays the reference-layout one (safe_io.py:361 loads float32)
    measure('configs3_hypergeometric_uint8_matrix', safepy_amd.LayoutGraph(workloads.uniform_layout(4, n)), 'euclidean', b.astype(np.uint8), {}, 3)
    return out


def example3_extra(np, cpu_leg):
    """The reference's only published number (examples/Example_3_Scatterplot_annotation.ipynb:73,104,147-153: 16 s for
    compute_pvalues(num_permutations=10000) on the 1586-node YeastPhenome UMAP scatter, one quantitative attribute, hardware
    not stated) on a surrogate of that shape, through the same calls: load_network(.scatter) -> define_neighborhoods('euclidean',
    0.06) -> load_attributes(DataFrame) -> compute_pvalues(num_permutations=10000); seeded (NumPy-compatible stream) and with
    the notebook's default random_seed=None.  The oracle's full run is timed beside it when the CPU leg is on."""
    import tempfile
    import logging
    import safepy_amd
    from safepy_amd import workloads
    out = {'published': {'seconds': 16.0, 'permutations_per_s': 599.8, 'source': 'Example_3_Scatterplot_annotation.ipynb cell 11 (tqdm line), hardware not stated'}}
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'surrogate_UMAP_1586.scatter')
        keys, xy, att = workloads.example3_scatter(path)
        logging.disable(logging.WARNING)
        for seed, tag in ((0, 'seeded'), (None, 'unseeded_default')):
            best = None
            for rep in range(3):
                sf = safepy_amd.SAFE(verbose=False)
                sf.random_seed = seed
                t = [time.perf_counter()]
                sf.load_network(network_file=path, node_key_attribute='key')
                t.append(time.perf_counter())
                sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.06)
                t.append(time.perf_counter())
                sf.load_attributes(attribute_file=att)
                t.append(time.perf_counter())
                sf.compute_pvalues(num_permutations=10000)
                nes = np.asarray(sf.nes)
                t.append(time.perf_counter())
                row = {'load_network_ms': 1e3 * (t[1] - t[0]), 'define_neighborhoods_ms': 1e3 * (t[2] - t[1]), 'load_attributes_ms': 1e3 * (t[3] - t[2]),
                       'compute_pvalues_and_read_nes_ms': 1e3 * (t[4] - t[3]), 'enriched_neighborhoods': int(np.asarray(sf.nes_binary).sum())}
                if best is None or row['compute_pvalues_and_read_nes_ms'] < best['compute_pvalues_and_read_nes_ms']:
                    best = row
                del sf, nes
            out[tag] = best
        logging.disable(logging.NOTSET)
        if cpu_leg:
            from oracle import safe_oracle as orc
            a = orc.neighborhoods_euclidean(xy, 0.06)
            b = att.to_numpy(dtype=np.float64)
            sample = 2000                                      # a bounded sample (~20 s): the loop is linear in the permutation count
            t0 = time.perf_counter()
            orc.compute_pvalues(a, b.copy(), enrichment_type='auto', num_permutations=sample, random_seed=0)
            cpu_s = (time.perf_counter() - t0) * 10000.0 / sample
            out['cpu_oracle'] = {'compute_pvalues_s': cpu_s, 'cores': effective_cores(), 'kind': 'port',
                                 'sample': '%d of the 10000 permutations of the same call (1586 nodes x 1 attribute), scaled linearly; NumPy/SciPy oracle' % sample}
            out['speedup_vs_cpu_oracle_seeded'] = cpu_s / (1e-3 * out['seeded']['compute_pvalues_and_read_nes_ms'])
            out['speedup_vs_published_seeded'] = 16.0 / (1e-3 * out['seeded']['compute_pvalues_and_read_nes_ms'])
    out['workload'] = '1586-node clustered scatter surrogate (safe-data is not available offline), euclidean r=0.06, 1 quantitative attribute (12 % NaN), 10000 permutations'
    return out


MFMA_I8_PEAK_TOPS = 5000.0     # MI355X_MICROARCH.md: dense i8 MFMA = 2x the 2.5 PF bf16 rate


def mfma_kernel(ctx, np, be):
    """The matrix-core form of the permutation test at ONE RANK'S SHARE of BASELINE.json configs[4], at its own size:
    N = 20 000 uniform layout, euclidean r = 0.1, 6250 quantitative f64 attributes x 1000 permutations (the whole
    compute_pvalues_by_randomization call on HBM-resident inputs, measured -- not extrapolated).
    Algorithmic work = the block-sparse GEMM the kernel runs: stored 256 x 32 membership blocks x 32-column tiles x
    i8 slices of the call x (permutations + 1 observed pass) x 2 ops per MAC."""
    from safepy_amd import workloads
    n, m, nperm = 20000, 6250, 1000
    xy = workloads.uniform_layout(4, n)
    nbr = be.Neighborhoods.euclidean(ctx, xy, 0.1 * (xy[:, 0].max() - xy[:, 0].min()))
    b = workloads.quantitative_attributes(3, n, m)
    attr = be.Attributes.from_host(ctx, b)
    del b
    outs = [ctx.alloc_f64(n, m) for _ in range(5)] + [ctx.alloc_f64(m)]
    res = None
    for _ in range(2):                                   # first call builds the block structure
        perms = be.Permutations(ctx, n, attr.row_flags(), nperm, 0)
        ctx.sync()
        t0 = time.perf_counter()
        be.randomization(ctx, nbr, attr, perms, 'sum', 'both', 0.05, [o.ptr for o in outs])
        ctx.sync()
        dt = time.perf_counter() - t0
        perms.close()
        name, ms, launches = ctx.last_kernel()
        res = (name, ms, launches, dt)
    name, ms, launches, dt = res
    blocks, pieces = be.block_count(nbr), be.piece_count(nbr)
    slices = be.last_mfma_slices(ctx)
    core, undecided = be.last_mfma_filter(ctx)
    # EXECUTED operations: the kernel issues MFMAs only for the 32 x 32 pieces of the stored blocks that hold a member
    # (safe_nbr_piece_count); the stored-block figure (what rounds 1-3 reported) is kept beside it.  The filtered form (round 5)
    # multiplies only the three high digits (core = 3) -- the compares they cannot decide are settled exactly from the low
    # digits by k_mfma_resolve -- so the same counts cost half the operations: `frac` prices the operations really executed,
    # `six_slice_equivalent_frac` what the round-4 kernel would have needed for the same result in the same time
    passes = (nperm + 2) if core < slices else (nperm + 1)
    ops = 2.0 * pieces * 32 * 32 * (32 * ((m + 31) // 32)) * core * passes
    stored_ops = 2.0 * blocks * 256 * 32 * (32 * ((m + 31) // 32)) * core * passes
    six_ops = 2.0 * pieces * 32 * 32 * (32 * ((m + 31) // 32)) * slices * (nperm + 1)
    tops = ops / (ms * launches * 1e-3) / 1e12
    out = {name: {'bound': 'mfma', 'workload': 'configs[4], one rank of 8: N=%d x M=%d quantitative f64 attributes x %d permutations, '
                                              '%d members per neighborhood on average' % (n, m, nperm, int(nbr.nnz / n)),
                  'kernel_ms': ms * launches, 'call_ms': 1e3 * dt, 'algorithmic_ops': ops, 'achieved': tops,
                  'peak': MFMA_I8_PEAK_TOPS, 'unit': 'TOP/s', 'frac': tops / MFMA_I8_PEAK_TOPS, 'i8_slices': slices,
                  'i8_slices_multiplied': core, 'compares_resolved_from_low_digits': undecided,
                  'compares_resolved_frac': (abs(undecided) / (float(n) * m * nperm)) if undecided else 0.0,
                  'six_slice_equivalent_frac': six_ops / (ms * launches * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS,
                  'membership_blocks_256x32': blocks, 'block_fill': nbr.nnz / (blocks * 256.0 * 32.0),
                  'pieces_32x32_multiplied': pieces, 'pieces_skipped_frac': 1.0 - pieces / (8.0 * blocks),
                  'stored_block_ops': stored_ops, 'stored_block_frac_of_peak': stored_ops / (ms * launches * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS,
                  'mfma_pipe_busy_pmc': mfma_pipe_busy(name),
                  # `achieved` / `frac` count the pieces really multiplied; the USEFUL rate counts one multiply-add per membership entry,
                  # column, permutation and slice
                  'useful_ops': 2.0 * float(nbr.nnz) * m * core * passes,
                  'useful_frac_of_peak': 2.0 * float(nbr.nnz) * m * core * passes / (ms * launches * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS,
                  'enrichments_per_s': float(n) * m * nperm / dt,
                  'config5_rank_share_seconds': dt}}
    # the same share with neighborhood_score_type='z-score' (safe_extras.py:19-31): 16-column tiles carrying value digits,
    # square digits and the not-NaN slice (7 i8 slices), f64 score evaluation on the exact sums in the epilogue
    perms = be.Permutations(ctx, n, attr.row_flags(), nperm, 0)
    ctx.sync()
    t0 = time.perf_counter()
    be.randomization(ctx, nbr, attr, perms, 'z-score', 'both', 0.05, [o.ptr for o in outs])
    ctx.sync()
    dtz = time.perf_counter() - t0
    perms.close()
    zname, zms, zlaunches = ctx.last_kernel()
    zcore, zund = be.last_mfma_filter(ctx)                # filtered: 3 high value | square slices + the not-NaN slice = 4 of 7
    zpasses = nperm if zcore < 7 else nperm + 1           # (the observed pass of the filtered form runs all seven slices once: counted below)
    zops = 2.0 * pieces * 32 * 32 * (32 * ((m + 15) // 16)) * (zcore * zpasses + (7 if zcore < 7 else 0))      # executed pieces, as above
    zstored = 2.0 * blocks * 256 * 32 * (32 * ((m + 15) // 16)) * (zcore * zpasses + (7 if zcore < 7 else 0))
    out[zname + '<z-score>'] = {'bound': 'mfma', 'workload': 'the same share, z-scores', 'kernel_ms': zms * zlaunches, 'call_ms': 1e3 * dtz,
                                'algorithmic_ops': zops, 'achieved': zops / (zms * zlaunches * 1e-3) / 1e12, 'peak': MFMA_I8_PEAK_TOPS,
                                'unit': 'TOP/s', 'frac': zops / (zms * zlaunches * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS, 'i8_slices': 7,
                                'i8_slices_multiplied': zcore, 'compares_resolved_from_low_digits': zund, 'config5_rank_share_seconds': dtz,
                                'stored_block_frac_of_peak': zstored / (zms * zlaunches * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS,
                                'mfma_pipe_busy_pmc': mfma_pipe_busy(zname), 'enrichments_per_s': float(n) * m * nperm / dtz}
    for o in outs:
        o.free()
    attr.close()
    nbr.close()
    return out


def off_fast_path(ctx, np, be, headline):
    """Binary randomization on the shapes beyond the blocked bit-sliced kernel (its 16-bit LDS offsets: 8 (N + 1) < 65536,
    N <= 8190) -- the reference has no such cliff (safe_extras.py:56-66 is one dgemm at any N).  Outside the timed region;
    unseeded tables (generated on the device), so the figures are the kernels':
      * N = 8300, the configs[1] surrogate's recipe at that size
      * configs[3]'s network: N = 20 000 uniform layout, euclidean r = 0.1, 577 members on average
    each x 1000 permutations (since round 6 both run k_permtest_bits_pre with sixteen-wave workgroups; round 5: k_permtest_bits
    with one wave per SIMD and the two-slice matrix-core form); cost per member-word = time / (membership entries x 64-attribute
    words x permutations), beside the headline kernel's (`headline`: kernel busy ms, nnz, words)."""
    import safepy_amd
    from safepy_amd import workloads
    out = {}
    cases = []
    d = workloads.costanzo_surrogate(seed=1, n=8300, m=2048, target_edges=int(28202 * 8300 / 3971), n_nan_rows=int(182 * 8300 / 3971))
    sf = safepy_amd.SAFE(verbose=False)
    sf.graph = safepy_amd.LayoutGraph(d['xy'], d['edge_u'], d['edge_v'], length=d['length'])
    sf.define_neighborhoods()
    cases.append(('N=8300 x M=2048 binary, shortpath_weighted_layout r=0.1 (the surrogate recipe)', sf._nbr, d['attributes'], sf))
    xy = workloads.uniform_layout(4, 20000)
    nbr20 = be.Neighborhoods.euclidean(ctx, xy, 0.1 * (xy[:, 0].max() - xy[:, 0].min()))
    rng = np.random.default_rng(5)
    cases.append(('configs[3] network: N=20000 euclidean r=0.1 x M=2048 binary (density 1 %)', nbr20,
                  np.asfortranarray((rng.uniform(size=(20000, 2048)) < 0.01).astype(np.float32)), None))
    nperm = 1000
    for what, nbr, b, keep in cases:
        n, m = b.shape
        attr = be.Attributes.from_host(ctx, b)
        outs = [ctx.alloc_f64(n, m) for _ in range(5)] + [ctx.alloc_f64(m)]
        best = None
        for _ in range(3):
            perms = be.Permutations(ctx, n, attr.row_flags(), nperm, None)
            ctx.sync()
            t0 = time.perf_counter()
            be.randomization(ctx, nbr, attr, perms, 'sum', 'both', 0.05, [o.ptr for o in outs])
            ctx.sync()
            dt = time.perf_counter() - t0
            perms.close()
            name, ms, launches = ctx.last_kernel()
            if best is None or dt < best[0]:
                best = (dt, name, ms * launches)
        dt, name, kms = best
        words = (m + 63) // 64
        out[what] = {'kernel': name, 'call_ms': 1e3 * dt, 'kernel_ms_sum': kms, 'enrichments_per_s': float(n) * m * nperm / dt,
                     'membership_nnz': int(nbr.nnz), 'members_mean': nbr.nnz / float(n),
                     'ps_per_member_word_permutation': 1e12 * dt / (float(nbr.nnz) * words * nperm)}
        for o in outs:
            o.free()
        attr.close()
        if keep is None:
            nbr.close()
    out['headline_for_comparison'] = {'kernel': 'k_permtest_bits_blk', 'kernel_busy_ms': headline['busy_ms'],
                                      'ps_per_member_word_permutation': 1e12 * 1e-3 * headline['busy_ms'] / (float(headline['nnz']) * headline['words'] * headline['perms'])}
    return out


def launch_ranks(n_gpus):
    """`python bench.py --gpus N` without a launcher: start `torch.distributed.run` with N ranks of this same script as a
    CHILD process (nothing here has touched a GPU yet -- no exec from a GPU-initialised process), pass its output through
    and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault('OMP_NUM_THREADS', '1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


class Workload:
    """Inputs of one bench mode, resident in HBM: membership handle, this rank's attribute block, the call's parameters."""

    def __init__(self, args, kind, scaling, perms, rank, world, local_rank, ctx, torch, np):
        import safepy_amd
        from safepy_amd import workloads, sharding
        self.kind, self.scaling, self.P = kind, scaling, int(perms)
        sf = safepy_amd.SAFE(verbose=False, device=local_rank)
        if kind == 'cfg4':
            n, m = 20000, 6250
            xy = workloads.uniform_layout(4, n)
            sf.graph = safepy_amd.LayoutGraph(xy)
            sf.define_neighborhoods(node_distance_metric='euclidean', neighborhood_radius=0.1)
            b = workloads.quantitative_attributes(3 + 100 * rank, n, m)          # every rank its own 6250 columns of the 50 000
            self.m_total = m * world
            self.dtype, self.order = np.float64, 'C'
            self.b_dev = torch.from_numpy(b).to('cuda')
            self.b_host = None
            self.name = ('configs[4]: 20000-node uniform layout, euclidean r=0.1, %d quantitative f64 attributes per rank x %d '
                         'permutations' % (m, self.P))
        else:
            data = workloads.costanzo_surrogate(seed=0, n=args.nodes, m=args.attrs, target_edges=int(28202 * args.nodes / 3971))
            sf.graph = safepy_amd.LayoutGraph(data['xy'], data['edge_u'], data['edge_v'], length=data['length'])
            sf.define_neighborhoods(node_distance_metric=args.metric, neighborhood_radius=args.radius)
            b = data['attributes']
            if scaling == 'strong':                    # ONE matrix, its columns split over the ranks (np.array_split, safe.py:1339)
                self.m_total = b.shape[1]
                c0, c1 = sharding.column_shards(self.m_total, world)[rank]
                b = np.asfortranarray(b[:, c0:c1])
            else:                                      # weak: every rank its own attribute block of the same shape
                if rank > 0:
                    b = workloads.go_like_binary(np.random.default_rng(1000 + rank), args.nodes, args.attrs, int(182 * args.nodes / 3971))
                self.m_total = b.shape[1] * world
            self.dtype, self.order = np.float32, 'F'
            self.b_dev = torch.from_numpy(np.ascontiguousarray(b.T)).to('cuda')      # F-order [n,m] == C-order [m,n]
            self.b_host = b
            self.name = ('configs[%d]: Costanzo-2016-shaped surrogate, %d nodes x %d GO-BP-like binary attributes x %d permutations, '
                         'metric %s r=%g, seed 0' % (2 if (scaling == 'strong' and self.P == 10000) else 1, b.shape[0],
                                                     self.m_total if scaling == 'strong' else b.shape[1], self.P, args.metric, args.radius))
        self.sf = sf
        self.nbr = sf._nbr
        self.n, self.m = b.shape
        self.counts = self.nbr.row_counts()
        self.units_per_step = float(self.n) * self.m_total * self.P          # node-attribute enrichments per step, all ranks


class StepProbe:
    """What happened to the process during each timed step, sampled after the step (a few microseconds): minor / major page
    faults, involuntary context switches (getrusage), CFS throttling of the container (cgroup cpu.stat nr_throttled /
    throttled_usec), device allocations made by the library (safe_alloc_count).  report() keeps the totals and the rows of the
    slowest steps, so that an outlier in `step_ms` comes with its cause."""

    def __init__(self):
        import resource
        self._resource = resource
        self._cg = None
        for path in ('/sys/fs/cgroup/cpu.stat', '/sys/fs/cgroup/cpu/cpu.stat'):
            if os.path.exists(path):
                self._cg = path
                break
        self.rows = []
        self._smaps0 = smaps_rss() if os.environ.get('SAFE_BENCH_SMAPS') == '1' else None      # (diagnostics: which mapping grew)
        self._thread_faults0 = thread_minor_faults()
        self._last = self._read()

    def _read(self):
        ru = self._resource.getrusage(self._resource.RUSAGE_SELF)
        thr, thr_us = 0, 0
        if self._cg:
            try:
                with open(self._cg) as f:
                    for line in f:
                        kv = line.split()
                        if kv[0] == 'nr_throttled':
                            thr = int(kv[1])
                        elif kv[0] in ('throttled_usec', 'throttled_time'):
                            thr_us = int(kv[1]) // (1000 if kv[0] == 'throttled_time' else 1)
            except (OSError, ValueError, IndexError):
                pass
        allocs = 0
        try:
            from safepy_amd import backend as be
            allocs = be.device_alloc_count()
        except Exception:
            pass
        rss = 0
        try:
            with open('/proc/self/statm') as f:
                rss = int(f.read().split()[1])                      # resident pages (a few microseconds)
        except (OSError, ValueError, IndexError):
            pass
        return (ru.ru_minflt, ru.ru_majflt, ru.ru_nivcsw, ru.ru_nvcsw, thr, thr_us, allocs, rss)

    def rebase(self):
        """Start counting from now (called right before the timed region)."""
        self.rows = []
        self._thread_faults0 = thread_minor_faults()
        self._last = self._read()

    def sample(self):
        now = self._read()
        self.rows.append(tuple(a - b for a, b in zip(now, self._last)))
        self._last = now

    def report(self, step_ms, timings=None):
        keys = ('minor_faults', 'major_faults', 'involuntary_switches', 'voluntary_switches', 'cfs_throttled_periods', 'cfs_throttled_us', 'device_allocations',
                'resident_pages_change')
        total = {k: int(sum(r[i] for r in self.rows)) for i, k in enumerate(keys)}
        order = sorted(range(len(step_ms)), key=lambda i: -step_ms[i])[:3]
        slow = [dict({'step': int(i), 'ms': float(step_ms[i])}, **{k: int(self.rows[i][j]) for j, k in enumerate(keys)}) for i in order if i < len(self.rows)]
        if timings and len(timings) == len(step_ms):            # where the step's own clocks put the time: host stream vs kernels
            for row in slow:
                t = timings[row['step']]
                row.update({'tables_enqueued_ms': t.get('tables_enqueued_ms'), 'draw_busy_ms': t.get('draw_busy_ms'),
                            'gpu_kernel_busy_ms': t.get('gpu_kernel_busy_ms'), 'gpu_kernel_sum_ms': t.get('gpu_kernel_ms')})
        # which threads took the minor faults of the timed region (two reads of /proc/self/task/*/stat, outside the steps)
        now = thread_minor_faults()
        by_thread = sorted(((v - self._thread_faults0.get(k, 0), k[1]) for k, v in now.items() if v - self._thread_faults0.get(k, 0) > 0), reverse=True)[:4]
        rep = {'totals_over_timed_steps': total, 'slowest_steps': slow, 'gc': 'frozen + disabled inside the timed region',
               'minor_faults_by_thread': [{'thread': name, 'minor_faults': int(cnt)} for cnt, name in by_thread]}
        if os.environ.get('SAFE_BENCH_SMAPS') == '2':              # nothing read before the timed region: the largest mappings after it
            after = smaps_rss()
            rep['mappings_largest_kb'] = [{'mapping': key, 'rss_kb': int(kb)} for kb, key in sorted(((kb, key) for key, kb in after.items()), reverse=True)[:14]]
        if self._smaps0 is not None:
            after = smaps_rss()
            grown = sorted(((kb - self._smaps0.get(key, 0), key) for key, kb in after.items() if kb - self._smaps0.get(key, 0) >= 1024), reverse=True)[:6]
            rep['mappings_grown_kb'] = [{'mapping': key, 'rss_kb_grown': int(kb)} for kb, key in grown]
        return rep


def smaps_rss():
    """Resident kB of every mapping of this process: {'start-end name': kB} (/proc/self/smaps; tens of milliseconds -- diagnostics only)."""
    out, key = {}, None
    try:
        with open('/proc/self/smaps') as f:
            for line in f:
                head = line.split()
                if head and '-' in head[0] and not head[0].endswith(':'):
                    key = head[0] + ' ' + (head[5] if len(head) > 5 else '[anon]') + ' ' + head[1]
                elif line.startswith('Rss:') and key is not None:
                    out[key] = int(head[1])
    except OSError:
        pass
    return out


def thread_minor_faults():
    """Minor page faults of every thread of this process so far, by (tid, name): /proc/self/task/*/stat field 10."""
    out = {}
    for tid in os.listdir('/proc/self/task'):
        try:
            with open('/proc/self/task/%s/stat' % tid) as f:
                text = f.read()
            name = text[text.index('(') + 1:text.rindex(')')]
            out[(int(tid), name)] = int(text[text.rindex(')') + 2:].split()[7])
        except (OSError, ValueError, IndexError):
            pass
    return out


def thread_cpu_ms():
    """CPU milliseconds (user + system) of every thread of this process so far, by (tid, name): /proc/self/task/*/stat."""
    out = {}
    tick = os.sysconf('SC_CLK_TCK')
    for tid in os.listdir('/proc/self/task'):
        try:
            with open('/proc/self/task/%s/stat' % tid) as f:
                text = f.read()
            name = text[text.index('(') + 1:text.rindex(')')]
            rest = text[text.rindex(')') + 2:].split()
            out[(int(tid), name)] = 1e3 * (int(rest[11]) + int(rest[12])) / tick
        except (OSError, ValueError):
            pass
    return out


def run_mode(wl, n_steps, n_warmup, ctx, dist, torch, np, be, sharding, world, diag_exchange=False, seed=0):
    """Warm-up + exactly n_steps timed steps of sharding.randomization_step on `wl` (barrier + synchronize on both sides, MAX over
    ranks); returns the numbers of the mode.  A step = one compute_pvalues pass of this rank's block: statistics for the
    dispatch rule, whole-matrix row flags (N > 1: one small all-gather), the seeded legacy stream (one per node) + table
    kernels, the permutation-test kernels with the fused p-value / NES / binarisation epilogue and -- N > 1 -- the all-gather
    of every rank's result over RCCL / xGMI."""
    n, m, P = wl.n, wl.m, wl.P
    out = {k: torch.empty((n, m), dtype=torch.float64, device='cuda') for k in sharding.RANDOMIZATION_OUTPUTS}
    enriched = torch.empty((m,), dtype=torch.float64, device='cuda')
    table = be.nes_table(P)
    timings = []

    def step(exchange=True):
        attr = be.Attributes.from_device(ctx, wl.b_dev.data_ptr(), wl.dtype, n, m, order=wl.order)
        t = {}
        try:
            # seed 0: the NumPy-compatible seeded stream (BASELINE's configuration); None: an unseeded call, tables generated on the device
            sharding.randomization_step(ctx, wl.nbr, attr, wl.m_total, P, seed, out, enriched, 'sum', 'both', 0.05,
                                        table=table, exchange=exchange, timing=t)
            timings.append(t)
        finally:
            attr.close()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(k, fn, probe=None):
        fence()
        t_begin = time.perf_counter()
        per_step = []
        for _ in range(k):
            ts = time.perf_counter()
            fn()
            per_step.append(1e3 * (time.perf_counter() - ts))     # (a step returns after its own stream synchronisation)
            if probe is not None:
                probe.sample()
        fence()
        seconds = time.perf_counter() - t_begin
        if dist is not None:                                       # the slowest rank's clock
            t = torch.tensor([seconds], dtype=torch.float64, device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            seconds = float(t.item())
        return seconds, per_step

    # measurement hygiene: no cyclic-GC pass inside the timed region (a generation-2 collection over torch's and NumPy's
    # objects is a multi-millisecond pause in a 3 ms step); what the OS did to each step is recorded beside its duration.
    # The collection runs BEFORE the last warm-up steps: the step right after it re-faults what the allocator gave back
    # (seen as ~3800 minor faults and a 10 ms step at the start of the timed region).
    import gc
    n_tail = min(n_warmup, 5)
    for _ in range(n_warmup - n_tail):
        step()
    gc.collect()
    gc.freeze()
    gc_was_enabled = gc.isenabled()
    gc.disable()
    probe = StepProbe()                 # (its imports and file look-ups happen before the last warm-up steps, not between them and the timed region)
    for _ in range(n_tail):
        step()
    timings.clear()
    probe.rebase()
    cpu0 = time.process_time()
    try:
        elapsed, step_ms = timed(n_steps, step, probe)
    finally:
        if gc_was_enabled:
            gc.enable()
        gc.unfreeze()
    res = {'elapsed': elapsed, 'step_ms': step_ms, 'host_cpu_ms': 1e3 * (time.process_time() - cpu0) / n_steps,
           'timings': list(timings), 'kernel': ctx.last_kernel(), 'out': out, 'step_probe': probe.report(step_ms, timings)}
    mean = lambda key: float(np.mean([t.get(key, 0.0) for t in timings])) if timings else 0.0      # noqa: E731
    mine = {'role': timings[-1].get('role', 'own'), 'host_stream_ms': mean('tables_enqueued_ms'), 'draw_busy_ms': mean('draw_busy_ms'),
            'waited_for_producer_ms': mean('waited_for_producer_ms'), 'gpu_kernel_ms': mean('gpu_kernel_ms'), 'gpu_kernel_busy_ms': mean('gpu_kernel_busy_ms'),
            'exchange_ms': mean('exchange_ms'), 'host_cpu_ms_per_step': res['host_cpu_ms']}
    res['exchange_form'] = timings[-1].get('exchange') if timings else None
    if dist is not None:
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        res['per_rank'] = everyone
    else:
        res['per_rank'] = [mine]
    # the same step with the exchange replaced by a D2H copy of the rank's own NES block (what a host that only
    # wants the results on disk needs), and with no exchange at all -- diagnostics outside the timed region, N > 1 only
    if dist is not None and diag_exchange:
        k_diag = max(2, min(n_steps, 5))
        host_nes = torch.empty((n, m), dtype=torch.float64).pin_memory()

        def step_d2h():
            step(exchange=False)
            host_nes.copy_(out['nes'], non_blocking=True)
            torch.cuda.current_stream().synchronize()

        t_d2h, _ = timed(k_diag, step_d2h)
        k_cpu = 200 if os.environ.get('SAFE_BENCH_THREAD_CPU') == '1' else k_diag       # (10 ms clock ticks: many steps for per-thread figures)
        cpu1, thr1 = time.process_time(), thread_cpu_ms()
        t_none, _ = timed(k_cpu, lambda: step(exchange=False))
        cpu_none = [1e3 * (time.process_time() - cpu1) / k_cpu]
        if os.environ.get('SAFE_BENCH_THREAD_CPU') == '1':
            thr2 = thread_cpu_ms()
            busy = sorted(((thr2[k] - thr1.get(k, 0.0)) / k_cpu, k[1], k[0]) for k in thr2)
            sys.stderr.write('rank %d threads, CPU ms per step (step %.2f ms): %s\n' % (
                dist.get_rank(), 1e3 * t_none / k_cpu, ', '.join('%s[%d] %.2f' % (nm, tid, ms) for ms, nm, tid in busy[::-1] if ms > 0.02)))
        t_none, k_diag_none = t_none * k_diag / k_cpu, k_diag
        everyone = [None] * world
        dist.all_gather_object(everyone, cpu_none[0])
        form = res['exchange_form'] or {}
        res['exchange_report'] = {'all_gather_ms_per_step': 1e3 * elapsed / n_steps, 'd2h_only_ms_per_step': 1e3 * t_d2h / k_diag,
                                  'no_exchange_ms_per_step': 1e3 * t_none / k_diag, 'steps_timed': k_diag,
                                  'host_cpu_ms_per_step_no_exchange_per_rank': everyone,
                                  'form': form.get('form'), 'bytes_received_per_rank': form.get('bytes_received'),
                                  'd2h_bytes_per_rank': int(8 * n * m)}
    return res


def roofline_of(wl, res, ctx, np, be):
    """The `roofline` object of the mode's dominant kernel: algorithmic bytes (or ops) per launch over the launch's HIP-event
    duration.  For the bit-sliced kernel the BINDING resources (VALU issue + LDS gather) come first: it is neither HBM- nor
    MFMA-bound, the nominal HBM line is kept because the contract asks for hbm | mfma."""
    n, m, P = wl.n, wl.m, wl.P
    kname, _, launches = res['kernel']
    launches = max(int(launches), 1)
    k_total = float(np.mean([t['gpu_kernel_ms'] for t in res['timings']]))
    k_ms = k_total / launches                               # average duration of ONE launch (HIP events)
    # consecutive launches run on two streams and overlap: launches x k_ms exceeds the time the GPU spent on them (and can exceed
    # the step); the union of the launches' intervals is what compares with ms_per_step
    k_busy = float(np.mean([t.get('gpu_kernel_busy_ms', 0.0) for t in res['timings']]))
    span = int(np.ceil(P / launches))                       # permutations per launch
    if kname.startswith('k_permtest_mfma'):
        blocks, slices = be.block_count(wl.nbr), be.last_mfma_slices(ctx)
        core, undecided = be.last_mfma_filter(ctx)
        # EXECUTED slices: the filtered form multiplies only the high digits (core = 3 of 6) over the permutations and forms the
        # observed score in two extra passes of three slices; the six-slice form carries the observed pass in every launch
        passes = (P + 2) if core < slices else (P + 1)
        ops = 2.0 * be.piece_count(wl.nbr) * 32 * 32 * (32 * ((m + 31) // 32)) * core * passes / launches      # executed pieces only
        useful = 2.0 * float(wl.nbr.nnz) * m * core * passes / launches
        tops = ops / (k_ms * 1e-3) / 1e12
        return {'bound': 'mfma', 'kernel': kname, 'achieved': tops, 'peak': MFMA_I8_PEAK_TOPS, 'unit': 'TOP/s', 'frac': tops / MFMA_I8_PEAK_TOPS,
                'traffic': None, 'kernel_ms': k_ms, 'launches_per_step': launches, 'kernel_busy_ms_per_step': k_busy,
                'algorithmic_ops': ops, 'i8_slices': slices, 'i8_slices_multiplied': core, 'compares_resolved_from_low_digits': undecided,
                'mfma_pipe_busy_pmc': mfma_pipe_busy(kname),
                'useful_mac_frac': useful / (k_ms * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS,
                'block_fill': float(wl.nbr.nnz) / (blocks * 256.0 * 32.0)}
    # Algorithmic HBM bytes of ONE launch: SURVEY 8(d) compulsory traffic = one read of the
    # attribute block, the permutation rows it consumes, the membership, one read-modify-write of its counters.
    n_wg = -(-m // 64)
    n_pad = -(-n // 64) * 64
    nnz = int(wl.nbr.nnz)
    if kname in ('k_permtest_bits_pre', 'k_permtest_bits_blk'):
        alg_bytes = 8 * (n + 1) * n_wg + 2 * nnz * span + 2 * nnz + 2 * 4 * n_pad * m
    elif kname == 'k_permtest_bits':
        alg_bytes = 8 * (n + 1) * n_wg + 2 * (n + 8) * span + 2 * nnz + 2 * 4 * n_pad * m
    else:
        alg_bytes = n * m * 4 + P * (n + 1) * 4 + nnz * 4 + 5 * n * m * 8
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    traffic, traffic_stamp = pmc_traffic(kname, with_stamp=True)
    binding = binding_resources(ctx.num_cu, kname, k_busy)
    roof = {}
    if binding is not None and kname.startswith('k_permtest_bits'):
        # what actually bounds the kernel (PMC passes of this same command, profiles/): the busier of its two pipes
        roof.update({'binding_resource': 'VALU issue (four in-order waves per SIMD)', 'binding_unit': 'fraction of the measured issue peak, over the time the kernel is running',
                     'binding_frac': binding.get('valu_issue_frac_of_measured_peak_while_busy', max(binding['valu_issue_frac'], binding['lds_pipe_busy_frac'])),
                     'binding_resource_utilisation': binding})
    roof.update({'bound': 'hbm', 'kernel': kname, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_stamp, 'kernel_ms': k_ms,
                 'launches_per_step': launches, 'kernel_busy_ms_per_step': k_busy,
                 'launch_overlap_note': 'launches alternate between two streams and overlap: launches_per_step x kernel_ms (sum of '
                                        'durations) exceeds kernel_busy_ms_per_step (union of their intervals)',
                 'permutations_per_launch': span, 'algorithmic_bytes': alg_bytes,
                 'note': 'nominal: the kernel is not HBM-bound (SURVEY 8d); DESIGN.md section 4',
                 'enrichments_per_s_kernel_only': float(n) * m * span / (k_ms * 1e-3)})
    return roof


def main():
    args = parse()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        args.gpus = world                          # under a launcher the launcher's world size is the truth

    # RCCL brings its own HIP streams: with the runtime's default of four hardware queues the context's two kernel streams and
    # its table stream end up sharing queues with them -- a one-rank RCCL group measured 5.1-5.3 ms per step against 4.4-4.6 with
    # eight queues (or with the context created before the group: both are done here)
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    import numpy as np
    import torch
    import safepy_amd                                               # noqa: F401
    from safepy_amd import backend as be
    from safepy_amd import sharding
    # host threads per rank: ranks of one node share the host -- sleeping host
    # waits when a rank has fewer than three cores to itself (before the context exists)
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', str(world)))
    host_cfg = be.configure_host_for_ranks(local_world)

    # SAFE_BENCH_SHARE_DEVICE=1 (diagnostics on a one-GPU box): every rank on device 0, exchange staged through gloo -- the same
    # host pipeline, shared permutation stream and integer exchange; the GPU itself is time-shared, so `value` means nothing
    share_device = os.environ.get('SAFE_BENCH_SHARE_DEVICE') == '1'
    if share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    numa_node = None if os.environ.get('SAFE_BENCH_NO_PIN') == '1' else be.pin_threads_to_device_numa(
        local_rank, reserve_draw_cores=int(os.environ.get('SAFE_BENCH_DRAW_CORES', '2')))
    torch.set_num_threads(1)      # no OpenMP spinning next to the host draw thread (container CPU quotas throttle it)
    ctx = be.Context.default(local_rank)          # (before the process group: see GPU_MAX_HW_QUEUES above)
    dist = None
    force_dist = os.environ.get('SAFE_BENCH_FORCE_DIST') == '1'      # exercise the collectives with a single rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        if force_dist and world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29533')
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        if share_device:
            dist.init_process_group(backend='gloo')
        else:
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    # ---------------- the timed mode: inputs (untimed) resident in HBM, then W warm-up and exactly K timed steps ----
    wl = Workload(args, args.workload, args.scaling, args.perms, rank, world, local_rank, ctx, torch, np)
    probe_unseeded = os.environ.get('SAFE_BENCH_SEED') == 'none'       # (probes only: the headline mode as an unseeded call)
    if probe_unseeded:
        wl.name += ' [SAFE_BENCH_SEED=none: random_seed=None, tables generated on the device -- NOT the headline configuration]'
    res = run_mode(wl, args.steps, args.warmup, ctx, dist, torch, np, be, sharding, world, diag_exchange=True,
                   seed=None if probe_unseeded else 0)

    line = None
    if rank == 0:
        ms_per_step = 1e3 * res['elapsed'] / args.steps
        value = wl.units_per_step / (res['elapsed'] / args.steps)
        kname = res['kernel'][0]
        roof = roofline_of(wl, res, ctx, np, be)
        line = {
            'metric': 'node-attribute enrichments/sec (nodes x attrs x perms), compute_pvalues permutation test',
            'value': value, 'unit': 'enrichments/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': wl.scaling, 'vs_baseline': None,
            'dtype': ('i8 (exact fixed-point slices on the matrix cores; f64 outputs)' if kname.startswith('k_permtest_mfma') else
                      'u1 (bit-sliced integer counts; f64 outputs)' if kname != 'k_permtest_gather' else 'f64'),
            'data': 'synthetic' if not share_device else 'synthetic (DIAGNOSTIC: all ranks share device 0, value is not a measurement)',
            'config': {'workload': wl.name, 'nodes': wl.n, 'attributes_per_gpu': wl.m, 'attributes_total': wl.m_total,
                       'permutations': wl.P, 'membership_nnz': int(wl.nbr.nnz),
                       'neighbors_per_node_mean': float(wl.counts.mean()), 'neighbors_per_node_std': float(wl.counts.std()),
                       'parallelism': 'attribute shards x%d (%s scaling)' % (world, wl.scaling)},
            'roofline': roof,
            'kernel_share_of_step': (roof.get('kernel_busy_ms_per_step') or roof['kernel_ms'] * roof['launches_per_step']) / ms_per_step,
            'per_rank': res['per_rank'],
            'host_cpu_ms_per_step': res['host_cpu_ms'], 'host_cores_usable': effective_cores(), 'host': host_cfg,
            'pinned_to_numa_node': numa_node,
            'step_ms_min_median_max': [float(np.min(res['step_ms'])), float(np.median(res['step_ms'])), float(np.max(res['step_ms']))],
            'step_ms_slowest3': [float(x) for x in sorted(res['step_ms'])[-3:]],
            'step_ms_deciles': [round(float(x), 3) for x in np.percentile(res['step_ms'], list(range(10, 100, 10)))],
            'step_ms_mean_by_quarter_of_run': [round(float(np.mean(q)), 3) for q in np.array_split(np.asarray(res['step_ms']), 4)],
            # the host draw chain's busy time of the steps in each decile of the step-time distribution (sorted by step time: a slow
            # chain thread shows as the last entries growing with the step)
            'draw_busy_ms_by_step_decile': _by_decile(res['step_ms'], [t.get('draw_busy_ms') for t in res['timings']]),
            # the first steps of a fresh process are slower: which clock moves (kernels' busy time / host stream / draw chain), in run order
            'first_8_steps_ms': [[round(float(res['step_ms'][i]), 3), round(float(res['timings'][i].get('gpu_kernel_busy_ms') or 0.0), 3),
                                  round(float(res['timings'][i].get('tables_enqueued_ms') or 0.0), 3), round(float(res['timings'][i].get('draw_busy_ms') or 0.0), 3)]
                                 for i in range(min(8, len(res['step_ms']), len(res['timings'])))],
            'draw_threads': {'placement': 'persistent per context',
                             'reserved_cores': int(os.environ.get('SAFE_BENCH_DRAW_CORES', '2')) if numa_node is not None else 0,
                             # the chain is drawn by two threads at once, the faster one publishes each chunk (rng.cpp, safe_perms::twin)
                             'twin_chain': bool(res['timings'][-1].get('twin_chain')),
                             'chunks_per_step': res['timings'][-1].get('chunks'),
                             'chunks_won_by_twin_per_step_mean': float(np.mean([t.get('chunks_won_by_twin') or 0 for t in res['timings']]))},
            'library_build': be.build_info(),
            # what paces the seeded step on THIS box: the host's masked-rejection chain (one thread; safe_extras.py:46,58) against the
            # kernels' busy time -- the step is the larger of the two plus the pipeline's fill and drain
            'draw_chain': {'cpu_model': _cpu_model(), 'path': be.build_info().rsplit('seeded draw chain: ', 1)[-1],
                           'us_per_permutation': 1e3 * float(np.mean([t.get('draw_busy_ms') or 0.0 for t in res['timings']])) / max(1, wl.P),
                           'busy_ms_per_step': float(np.mean([t.get('draw_busy_ms') or 0.0 for t in res['timings']])),
                           'step_minus_chain_ms': ms_per_step - float(np.mean([t.get('draw_busy_ms') or 0.0 for t in res['timings']])),
                           'step_minus_kernels_busy_ms': ms_per_step - float(np.mean([t.get('gpu_kernel_busy_ms') or 0.0 for t in res['timings']]))},
            'step_probe': res['step_probe'],
        }
        if 'exchange_report' in res:
            line['exchange'] = res['exchange_report']
    # the same workload as an UNSEEDED call (random_seed=None, the reference's default): tables generated on the device, no host
    # stream -- outside the timed region, reported beside the seeded headline
    if args.extras and args.workload == 'cfg1' and args.scaling == 'weak':
        unseeded = {}
        for perms_u, steps_u in ((wl.P, max(5, min(args.steps, 20))), (10000, 5)):
            wl_u = wl if perms_u == wl.P else Workload(args, 'cfg1', 'weak', perms_u, rank, world, local_rank, ctx, torch, np)
            r_u = run_mode(wl_u, steps_u, 2, ctx, dist, torch, np, be, sharding, world, seed=None)
            if rank == 0:
                unseeded['%d_permutations' % perms_u] = {
                    'ms_per_step': 1e3 * r_u['elapsed'] / steps_u, 'value': wl_u.units_per_step / (r_u['elapsed'] / steps_u),
                    'unit': 'enrichments/s', 'steps': steps_u, 'stream': r_u['per_rank'][0]['role'],
                    'gpu_kernel_ms': r_u['per_rank'][0]['gpu_kernel_ms'], 'host_cpu_ms_per_step': r_u['host_cpu_ms']}
            del r_u
            if wl_u is not wl:
                del wl_u
        if rank == 0:
            line['unseeded_device_stream'] = dict(unseeded, note='random_seed=None (the reference default: OS entropy, no stream to reproduce): '
                                                  'i.i.d. uniform permutations generated on the GPU (Philox4x32-10 + Fisher-Yates in LDS); '
                                                  'the seeded headline above reproduces NumPy\'s MT19937 stream on the host')
    a_dense = wl.sf.neighborhoods if (rank == 0 and args.cpu_perms > 0 and world == 1 and wl.kind == 'cfg1') else None
    b_host = wl.b_host
    del res

    # ---------------- N > 1: the other configurations BASELINE.json names for several GPUs, outside the timed region -----------
    if world > 1 and args.multi_extras and args.workload == 'cfg1' and args.scaling == 'weak':
        extras = {}
        wl2 = Workload(args, 'cfg1', 'strong', 10000, rank, world, local_rank, ctx, torch, np)
        r2 = run_mode(wl2, 3, 1, ctx, dist, torch, np, be, sharding, world)
        if rank == 0:
            extras['configs2_strong_scaling'] = {
                'workload': wl2.name + ', columns np.array_split over %d ranks' % world, 'scaling': 'strong', 'steps': 3,
                'ms_per_step': 1e3 * r2['elapsed'] / 3, 'value': wl2.units_per_step / (r2['elapsed'] / 3), 'unit': 'enrichments/s',
                'attributes_per_gpu': wl2.m, 'per_rank': r2['per_rank'], 'exchange': r2['exchange_form'],
                'amdahl_note': 'the permutation stream is sequential (one draw thread per node, ~1.7-2.2 ms per 1000 permutations): '
                               'a 10 000-permutation step cannot go below that thread\'s time, whatever the rank count'}
        r2u = run_mode(wl2, 3, 1, ctx, dist, torch, np, be, sharding, world, seed=None)
        if rank == 0:
            extras['configs2_strong_scaling_unseeded'] = {
                'workload': wl2.name + ', columns np.array_split over %d ranks, random_seed=None (tables generated on every rank\'s device)' % world,
                'scaling': 'strong', 'steps': 3, 'ms_per_step': 1e3 * r2u['elapsed'] / 3, 'value': wl2.units_per_step / (r2u['elapsed'] / 3),
                'unit': 'enrichments/s', 'attributes_per_gpu': wl2.m, 'per_rank': r2u['per_rank'], 'exchange': r2u['exchange_form']}
        del wl2, r2, r2u
        wl4 = Workload(args, 'cfg4', 'weak', 1000, rank, world, local_rank, ctx, torch, np)
        r4 = run_mode(wl4, 1, 1, ctx, dist, torch, np, be, sharding, world)
        if rank == 0:
            extras['configs4_rank_share'] = {
                'workload': wl4.name, 'scaling': 'weak', 'steps': 1, 'ms_per_step': 1e3 * r4['elapsed'],
                'value': wl4.units_per_step / r4['elapsed'], 'unit': 'enrichments/s', 'attributes_per_gpu': wl4.m,
                'per_rank': r4['per_rank'], 'exchange': r4['exchange_form'], 'roofline': roofline_of(wl4, r4, ctx, np, be)}
        del wl4, r4
        if rank == 0:
            line['multi_gpu_configs'] = extras

    if rank == 0:
        if a_dense is not None:                               # the CPU leg runs on rank 0 at N = 1 only
            line['cpu_baseline'] = cpu_baseline(a_dense, b_host, args.cpu_perms)
            line['cpu_baseline']['gpu_over_cpu'] = line['value'] / line['cpu_baseline']['value']      # mostly the algorithm (sparse bit slices vs dense dgemm), not kernel quality
        if args.extras and world == 1 and args.workload == 'cfg1':
            del wl
            line['hbm_bound_kernels'] = hbm_kernels(ctx, torch, np, be)
            line['mfma_bound_kernels'] = mfma_kernel(ctx, np, be)
            line['dropin_compute_pvalues'] = dropin_extras(np, torch)
            line['example3_published_shape'] = example3_extra(np, args.cpu_perms > 0)
            line['off_fast_path_shapes'] = off_fast_path(ctx, np, be, {'busy_ms': roof.get('kernel_busy_ms_per_step') or 0.0, 'nnz': line['config']['membership_nnz'],
                                                                         'words': (line['config']['attributes_per_gpu'] + 63) // 64, 'perms': line['config']['permutations']})
    # RCCL prints a version banner through C stdio (flushed at exit when stdout is a pipe): tear the group down and flush every
    # rank's C buffers first, so that rank 0's JSON line is the LAST line of the job's output
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        import ctypes
        ctypes.CDLL(None).fflush(None)
        if rank == 0 and world > 1:
            time.sleep(0.3)                                     # (the other ranks flush and leave)
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == '__main__':
    main()
