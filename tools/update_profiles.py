"""Copy the summaries of a tools/prof_round.sh run (gpurun_out/<tag>/) into profiles/: kernel statistics, FETCH_SIZE /
WRITE_SIZE sums and profiles/pmc_traffic.json (per-launch HBM bytes bench.py reports as roofline.traffic).
usage: update_profiles.py <tag>"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src_dir = os.path.join(ROOT, 'gpurun_out', tag)
src = open(os.path.join(src_dir, 'pmc_traffic.txt')).read().splitlines()
vals = {}
for line in src:
    m = re.match(r'(.{60}) (\S+)\s+sum=(\S+)\s+n=(\d+)', line)
    if m:
        vals[(m.group(1).strip(), m.group(2))] = (float(m.group(3)), int(m.group(4)))


def get(sub, ctr):
    for (k, c), v in vals.items():
        if sub in k and c == ctr:
            return v
    return None


out = {}


def entry(key, sub, corr=1.0, note=None):
    f, w = get(sub, 'FETCH_SIZE'), get(sub, 'WRITE_SIZE')
    if f is None or w is None:
        return
    e = {'fetch_bytes_per_launch': f[0] * 1024 / f[1], 'write_bytes_per_launch': w[0] * 1024 / w[1], 'fetch_correction': corr,
         'hbm_bytes_per_launch': corr * f[0] * 1024 / f[1] + w[0] * 1024 / w[1], 'launches_profiled': f[1]}
    if note:
        e['note'] = note
    out[key] = e


entry('k_permtest_bits_pre', 'k_permtest_bits_pre')
entry('k_euclid_dense', 'k_euclid_dense')
entry('k_hyp_emit', 'k_hyp_emit', 1.0,
      'reads are 12 B/lane count records + 16 B/lane table slabs (mostly L2 hits): the x2 correction of 16 B/lane streams is not '
      'applied (uncalibrated width); algorithmic bytes 5.21 GB (4.80 GB written + 0.41 GB of packed counts read); 4 % of the writes '
      'are repeated last rows of short row batches (1-2 % with the tail rule)')
entry('k_permtest_mfma<counts> (split form)', 'k_permtest_mfma<true, 6>', 2.0)
entry('k_mfma_planes01_rows', 'k_mfma_planes01_rows', 2.0)
entry('k_permute_cols', 'k_permute_cols')
entry('k_counts_finalize', 'k_counts_finalize')
out['_source'] = ('rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python3 bench.py --steps 1 --warmup 1 '
                  '--cpu-perms 0 (tools/prof_round.sh), round 1, MI355X; counter unit KiB; x2 gfx950 FETCH_SIZE correction applied only to '
                  'kernels whose reads are 16 B/lane streams (MI355X_MICROARCH.md, HBM section); see profiles/r01_pmc_traffic.txt')
json.dump(out, open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json'), 'w'), indent=1)
hdr = ('# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --steps 1 --warmup 1 --cpu-perms 0 '
       '(tools/prof_round.sh; round 1, MI355X); sums over launches, unit KiB\n')
open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.txt'), 'w').write(hdr + '\n'.join(src) + '\n')
ks = open(os.path.join(src_dir, 'kernel_stats.txt')).read()
line = open(os.path.join(src_dir, 'bench_line.json')).read().strip()
hdr = ('# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --cpu-perms 0 (tools/prof_round.sh; round 1, MI355X): '
       'headline steps + extras (K1 distance, K4 hypergeometric at 20000 x 10000, matrix-core permutation kernel)\n'
       '# bench line of the same run: ' + line[:700] + ' ...\n')
open(os.path.join(ROOT, 'profiles', 'r01_bench_kernel_stats.txt'), 'w').write(hdr + ks)
print({k: round(v['hbm_bytes_per_launch'] / 1e6, 1) for k, v in out.items() if isinstance(v, dict)})
