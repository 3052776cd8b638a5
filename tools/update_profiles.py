"""Turn the summaries of a tools/prof_round.sh run (gpurun_out/<tag>/) into the committed profiles of a round:
  profiles/rNN_bench_kernel_stats.txt   per-kernel durations of the default bench run (rocprofv3 --kernel-trace --stats)
  profiles/rNN_pmc_permtest_bits.txt    SQ / TCC counters of the headline step
  profiles/rNN_pmc_traffic.txt          FETCH_SIZE / WRITE_SIZE sums per kernel of the whole bench
  profiles/pmc_bits.json, pmc_traffic.json   the figures bench.py quotes as roofline.binding_resource_utilisation / traffic,
                                             stamped with the commit and the hash of the kernel sources they were taken at
                                             (bench.py marks them "stale" when the sources differ)
usage: update_profiles.py <tag> <round, e.g. r02>    -- run at the commit that was profiled"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                     # noqa: E402  (kernel_source_sha256)

tag, rnd = sys.argv[1], sys.argv[2]
src_dir = os.path.join(ROOT, 'gpurun_out', tag)
commit = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], cwd=ROOT, capture_output=True, text=True).stdout.strip()
stamp = {'commit': commit, 'kernel_source_sha256': bench.kernel_source_sha256()}


def counters(path):
    vals = {}
    for line in open(path).read().splitlines():
        m = re.match(r'(.{60}) (\S+)\s+sum=(\S+)\s+n=(\d+)', line)
        if m:
            vals[(m.group(1).strip(), m.group(2))] = (float(m.group(3)), int(m.group(4)))
    return vals


# ---- traffic per launch (counter unit KiB)
traffic = counters(os.path.join(src_dir, 'pmc_traffic.txt'))


def get(vals, sub, ctr):
    for (k, c), v in vals.items():
        if sub in k and c == ctr:
            return v
    return None


# ---- calibration of the two counters on this box (tools/pmc_calib.sh -> gpurun_out/calib/calib.txt): known bytes / reported
calib = {}
calib_path = os.path.join(ROOT, 'gpurun_out', 'calib', 'calib.txt')
if os.path.exists(calib_path):
    cv = counters(calib_path)
    known = 2.0 * (2 << 30) / 1024.0                       # two launches of 2 GiB each, in KiB
    for kern, ctr in (('k_read16', 'FETCH_SIZE'), ('k_read12', 'FETCH_SIZE'), ('k_read8', 'FETCH_SIZE'), ('k_read4', 'FETCH_SIZE'),
                      ('k_gather192', 'FETCH_SIZE'), ('k_write16', 'WRITE_SIZE'), ('k_atomic4', 'WRITE_SIZE'), ('k_atomic4', 'FETCH_SIZE')):
        v = get(cv, kern, ctr) if 'get' in globals() else None
        if v is None:
            for (k, c), vv in cv.items():
                if kern in k and c == ctr:
                    v = vv
        if v is not None:
            calib['%s:%s' % (kern, ctr)] = {'reported_KiB': v[0], 'requested_KiB': known,
                                            'bytes_per_reported_byte': (known / v[0]) if v[0] > 1e3 else None}
    calib['_reading'] = ('every coalesced read width (4 / 8 / 12 / 16 B per lane) reports exactly half of the bytes read, random 192-byte row '
                         'gathers two thirds of the bytes requested = half of the 128-byte lines touched: FETCH_SIZE x 2 = bytes that crossed '
                         'the fabric, for every kernel here; WRITE_SIZE is exact for 16 B/lane stores; a 4 B/lane atomicAdd counts its bytes '
                         'once as WRITE_SIZE and not at all as FETCH_SIZE (the read-modify-write happens memory-side)')
    json.dump(dict(calib, _stamp=stamp), open(os.path.join(ROOT, 'profiles', 'pmc_calibration.json'), 'w'), indent=1)
    open(os.path.join(ROOT, 'profiles', '%s_pmc_calibration.txt' % rnd), 'w').write(
        '# tools/pmc_calib.sh (tools/ubench/fetch_calib.hip: 2 launches x 2 GiB per kernel; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate '
        'passes; round %s, commit %s, MI355X); unit KiB\n' % (rnd, commit) + open(calib_path).read())

out = {}


FETCH_CORR = 2.0                                           # calibrated above: holds for every read pattern of these kernels


def entry(key, sub, corr=FETCH_CORR, note=None):
    f, w = get(traffic, sub, 'FETCH_SIZE'), get(traffic, sub, 'WRITE_SIZE')
    if f is None or w is None:
        return
    e = {'fetch_bytes_per_launch': f[0] * 1024 / f[1], 'write_bytes_per_launch': w[0] * 1024 / w[1], 'fetch_correction': corr,
         'hbm_bytes_per_launch': corr * f[0] * 1024 / f[1] + w[0] * 1024 / w[1], 'launches_profiled': f[1]}
    if note:
        e['note'] = note
    out[key] = e


entry('k_permtest_bits_blk (whole bench: launches of the unseeded and 10 000-permutation extras included)', 'k_permtest_bits_blk')
# the HEADLINE launches alone (tools/pmc_bits.sh passes 4 / 5: bench.py --extras 0): what roofline.traffic of the bench line quotes
_hb = os.path.join(src_dir, 'bits', 'pmc_summary.txt')
if os.path.exists(_hb):
    _mixed = traffic
    traffic = counters(_hb)
    entry('k_permtest_bits_blk', 'k_permtest_bits_blk', FETCH_CORR, 'headline launches only (bench.py --extras 0, warm-up + 1 step)')
    traffic = _mixed
entry('k_bits_observed', 'k_bits_observed')
entry('k_euclid_dense', 'k_euclid_dense')
entry('k_hyp_emit', 'k_hyp_emit', FETCH_CORR,
      'reads are 12 B/lane count records (calibrated: x2 like every coalesced width) + 16 B/lane table slabs (mostly L2 hits); '
      'algorithmic bytes 5.21 GB (4.80 GB written + 0.41 GB of packed counts read)')
entry('k_permtest_mfma<counts> (split form)', 'k_permtest_mfma<true, 6')
entry('k_permtest_mfma', 'k_permtest_mfma<false, 6')
entry('k_mfma_planes01_rows', 'k_mfma_planes01_rows')
entry('k_permute_cols', 'k_permute_cols')
entry('k_counts_finalize', 'k_counts_finalize')
out['_source'] = ('rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python3 bench.py --steps 1 --warmup 1 '
                  '--cpu-perms 0 (tools/prof_round.sh), round %s, MI355X; counter unit KiB; FETCH_SIZE x 2 for every kernel (gfx950 counts 128-byte '
                  'requests at 64 B: MI355X_MICROARCH.md, HBM section, re-measured for the access widths of these kernels in '
                  'profiles/pmc_calibration.json); Infinity-Cache hits are counted too: these are FABRIC bytes, an upper bound of the HBM bytes; '
                  'see profiles/%s_pmc_traffic.txt' % (rnd, rnd))
out['_stamp'] = stamp
json.dump(out, open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json'), 'w'), indent=1)
hdr = ('# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --steps 1 --warmup 1 --cpu-perms 0 '
       '(tools/prof_round.sh; round %s, commit %s, MI355X); sums over launches, unit KiB\n' % (rnd, commit))
open(os.path.join(ROOT, 'profiles', '%s_pmc_traffic.txt' % rnd), 'w').write(hdr + open(os.path.join(src_dir, 'pmc_traffic.txt')).read())

# ---- kernel statistics
ks = open(os.path.join(src_dir, 'kernel_stats.txt')).read()
line = open(os.path.join(src_dir, 'bench_line.json')).read().strip()
hdr = ('# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --cpu-perms 0 (tools/prof_round.sh; round %s, commit %s, '
       'MI355X): headline steps + extras (K1 distance, K4 hypergeometric at 20000 x 10000, matrix-core permutation kernel at one rank\'s '
       'configs[4] share)\n# bench line of the same run: %s ...\n' % (rnd, commit, line[:900]))
open(os.path.join(ROOT, 'profiles', '%s_bench_kernel_stats.txt' % rnd), 'w').write(hdr + ks)

hk = os.path.join(src_dir, 'kernel_stats_headline.txt')
if os.path.exists(hk):
    hline = open(os.path.join(src_dir, 'bench_line_headline.json')).read().strip()
    open(os.path.join(ROOT, 'profiles', '%s_bench_kernel_stats_headline.txt' % rnd), 'w').write(
        '# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --extras 0 --cpu-perms 0 (tools/prof_round.sh; round %s, commit %s, '
        'MI355X): the HEADLINE alone -- the average duration of k_permtest_bits_blk here is the kernel_ms of the bench line below '
        '(roofline.kernel_ms, HIP events); k_replay_targets / k_scan_* / k_permute_cols run beside it on the table streams\n'
        '# bench line of the same run: %s ...\n' % (rnd, commit, hline[:700]) + open(hk).read())
tl = os.path.join(src_dir, 'timeline.txt')
if os.path.exists(tl):
    open(os.path.join(ROOT, 'profiles', '%s_timeline_step_seeded.txt' % rnd), 'w').write(
        '# GPU timeline of one seeded headline step (tools/timeline_step.sh: rocprofv3 --kernel-trace -- python3 tools/trace_step.py; round %s, commit %s, '
        'MI355X).  columns: start, end (us from the first kernel), duration, gap to the previous kernel\'s end (negative = overlapped on '
        'another stream), kernel.  The first block is the first call of the process (statistics, bit planes, observed sums); the '
        'second block the steady step: upload + k_replay_targets on the replay stream, k_scan_* on the table stream, k_permute_cols + '
        'k_permtest_bits_blk alternating on two kernel streams.  A k_replay_targets / k_scan duration that spans a whole '
        'k_permtest_bits_blk launch is QUEUE WAIT (the persistent kernel holds the CUs; the table kernels get one as its workgroups '
        'retire), not work: alone, a replay of 128 permutations takes ~60 us\n' % (rnd, commit) + open(tl).read())

# ---- SQ / TCC counters of the headline kernel
bits_txt = open(os.path.join(src_dir, 'bits', 'pmc_summary.txt')).read()
hdr = ('# rocprofv3 --pmc <8 SQ counters | 8 SQ counters | TCC_HIT TCC_MISS GRBM_GUI_ACTIVE | FETCH_SIZE | WRITE_SIZE> --kernel-trace -- '
       'python3 bench.py --steps 1 --warmup 1 --cpu-perms 0 --extras 0 (tools/pmc_bits.sh: five separate passes; round %s, commit %s, '
       'MI355X); sums over the launches of 2 steps (warm-up + 1)\n' % (rnd, commit))
open(os.path.join(ROOT, 'profiles', '%s_pmc_permtest_bits.txt' % rnd), 'w').write(hdr + bits_txt)
bits = counters(os.path.join(src_dir, 'bits', 'pmc_summary.txt'))
kname = 'k_permtest_bits_blk'
doc = {'kernel': 'k_permtest_bits_blk<8, 0>', 'steps_profiled': 2}
for ctr in ('SQ_WAVE_CYCLES', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_INSTS_VALU', 'SQ_INSTS_LDS', 'SQ_LDS_IDX_ACTIVE',
            'SQ_LDS_BANK_CONFLICT', 'GRBM_GUI_ACTIVE', 'TCC_HIT_sum', 'TCC_MISS_sum'):
    v = get(bits, kname, ctr)
    if v is not None:
        doc[ctr] = v[0]
doc['valu_wave_insts_per_clock_per_simd_sustained'] = 0.43
doc['_source'] = 'profiles/%s_pmc_permtest_bits.txt (rocprofv3 --pmc, tools/pmc_bits.sh), tools/ubench/valu_issue.hip' % rnd
doc.update(stamp)
json.dump(doc, open(os.path.join(ROOT, 'profiles', 'pmc_bits.json'), 'w'), indent=1)
# ---- matrix-core kernel: pipe-busy fraction (tools/pmc_mfma.sh <tag>/mfma_sum sum ; <tag>/mfma_z z-score)
mf = dict(stamp)
txt_all = ''
for key, sub in (('mfma_pipe_busy_sum', 'mfma_sum'), ('mfma_pipe_busy_zscore', 'mfma_z')):
    path = os.path.join(src_dir, sub, 'pmc_summary.txt')
    if not os.path.exists(path):
        continue
    c = counters(path)
    # the kernel that carries the permutations: the filtered form's own kernel ('sum'), the four-slice z form; before round 5
    # (or with SAFE_HIP_MFMA_FILTER=0) the six- / seven-slice general kernel
    # (round 6: k_permtest_mfma_g<...> / k_permtest_mfma_gz<...>; the older names are kept for libraries run with SAFE_HIP_MFMA_FORM)
    want = [s for s in (('k_permtest_mfma_g<', 'k_permtest_mfma_f', 'k_permtest_mfma<false, 6') if sub == 'mfma_sum'
                        else ('k_permtest_mfma_gz<', 'k_permtest_mfma<false, 4, true', 'k_permtest_mfma<false, 7'))
            if get(c, s, 'SQ_VALU_MFMA_BUSY_CYCLES')]
    kern = want[0] if want else 'k_permtest_mfma'
    busy, gui, n_mfma = get(c, kern, 'SQ_VALU_MFMA_BUSY_CYCLES'), get(c, kern, 'GRBM_GUI_ACTIVE'), get(c, kern, 'SQ_INSTS_MFMA')
    mf[key + '_kernel'] = kern
    if busy and gui:
        # GRBM_GUI_ACTIVE sums the 8 XCDs' active clocks; the busy cycles sum over the 1024 SIMDs' matrix pipes
        mf[key] = busy[0] / (gui[0] / 8.0 * 1024.0)
        mf[key + '_counters'] = {'SQ_VALU_MFMA_BUSY_CYCLES': busy[0], 'GRBM_GUI_ACTIVE': gui[0], 'SQ_INSTS_MFMA': n_mfma[0] if n_mfma else None}
    txt_all += '## %s\n' % sub + open(path).read()
if txt_all:
    mf['_source'] = 'profiles/%s_pmc_permtest_mfma.txt (rocprofv3 --pmc, tools/pmc_mfma.sh: python3 tools/bench_big.py quant 1024 128 sum|z-score)' % rnd
    json.dump(mf, open(os.path.join(ROOT, 'profiles', 'pmc_mfma.json'), 'w'), indent=1)
    open(os.path.join(ROOT, 'profiles', '%s_pmc_permtest_mfma.txt' % rnd), 'w').write(
        '# rocprofv3 --pmc (four separate passes, --kernel-trace only) -- python3 tools/bench_big.py quant 1024 128 <sum|z-score> (tools/pmc_mfma.sh; round %s, '
        'commit %s, MI355X)\n' % (rnd, commit) + txt_all)
    print('mfma pipe busy', {k: round(v, 3) for k, v in mf.items() if k.startswith('mfma_pipe_busy') and isinstance(v, float)})
print({k: round(v['hbm_bytes_per_launch'] / 1e6, 1) for k, v in out.items() if isinstance(v, dict) and 'hbm_bytes_per_launch' in v})
print('LDS conflict share', doc.get('SQ_LDS_BANK_CONFLICT', 0) / max(doc.get('SQ_LDS_IDX_ACTIVE', 1), 1))
