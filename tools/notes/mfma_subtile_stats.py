"""How much of the matrix-core kernel's multiply work is multiplication by zero?  Rebuilds the block-sparse membership
of mfma.hip::build_blocks on the host (Hilbert order, 256-row groups x 32-column blocks) for the configs[4] layout and
counts, per stored block, the all-zero 32 x 32 sub-tiles (one per wave), and what a per-wave skip would leave on the
busiest SIMD of every super-step of 4 blocks (waves w and w + 4 share a SIMD)."""
import sys
import numpy as np
from scipy.spatial import cKDTree


def hilbert(ix, iy, order=16):
    d = np.zeros(ix.shape, dtype=np.uint64)
    x, y = ix.astype(np.int64).copy(), iy.astype(np.int64).copy()
    s = 1 << (order - 1)
    while s > 0:
        rx = ((x & s) > 0).astype(np.int64)
        ry = ((y & s) > 0).astype(np.int64)
        d += (np.uint64(s) * np.uint64(s) * ((3 * rx) ^ ry).astype(np.uint64))
        flip = (ry == 0) & (rx == 1)
        x = np.where(flip, s - 1 - (x & (s - 1)), x)    # (only the low bits matter from here on)
        y = np.where(flip, s - 1 - (y & (s - 1)), y)
        swap = ry == 0
        x, y = np.where(swap, y, x), np.where(swap, x, y)
        x &= s - 1
        y &= s - 1
        s >>= 1
    return d


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    xy = np.random.default_rng(4).uniform(size=(n, 2))
    r = 0.1 * (xy[:, 0].max() - xy[:, 0].min())
    span = max(np.ptp(xy[:, 0]), np.ptp(xy[:, 1]))
    key = hilbert(((xy[:, 0] - xy[:, 0].min()) / span * 65535).astype(np.int64), ((xy[:, 1] - xy[:, 1].min()) / span * 65535).astype(np.int64))
    order = np.argsort(key, kind='stable')
    pos = np.empty(n, dtype=np.int64)
    pos[order] = np.arange(n)
    tree = cKDTree(xy)
    pairs = tree.query_pairs(r, output_type='ndarray')          # strict '<' differs on a null set: statistics only
    rows = np.r_[pairs[:, 0], pairs[:, 1], np.arange(n)]
    cols = np.r_[pairs[:, 1], pairs[:, 0], np.arange(n)]
    u, p = pos[rows], pos[cols]
    nnz = len(u)
    # stored blocks: (group of 256 rows, block of 32 columns); sub-tiles: (32-row tile, 32-column block)
    blk = np.unique((u >> 8) * (1 << 20) + (p >> 5))
    sub = np.unique((u >> 5) * (1 << 20) + (p >> 5))
    n_blocks, n_sub = len(blk), len(sub)
    print('n=%d nnz=%d (%.0f per row); stored 256x32 blocks %d (fill %.3f); non-empty 32x32 sub-tiles %d of %d = %.3f (fill inside them %.3f)'
          % (n, nnz, nnz / n, n_blocks, nnz / (n_blocks * 8192.0), n_sub, n_blocks * 8, n_sub / (n_blocks * 8.0), nnz / (n_sub * 1024.0)))
    # per group: blocks in ascending column order, super-steps of 4; per super-step and wave the count of non-empty sub-tiles
    sub_set = set(sub.tolist())
    groups = {}
    for b in blk.tolist():
        groups.setdefault(b >> 20, []).append(b & ((1 << 20) - 1))
    tot_max_simd = tot_steps = tot_max_wave = 0
    tot_sorted = 0
    for g, kbs in groups.items():
        kbs.sort()
        occ = np.array([[((g * 8 + w) * (1 << 20) + kb) in sub_set for w in range(8)] for kb in kbs], dtype=np.int64)   # [block][wave]
        for variant in ('as stored', 'balanced'):
            o = occ
            if variant == 'balanced':
                # greedy: deal the blocks into super-steps so that the per-SIMD load of each stays level
                ns = -(-len(kbs) // 4)
                load = np.zeros((ns, 4), dtype=np.int64)
                fill = np.zeros(ns, dtype=np.int64)
                simd = occ[:, :4] + occ[:, 4:]
                for i in np.argsort(-simd.sum(axis=1), kind='stable'):
                    cand = [(int((load[s] + simd[i]).max()), int(fill[s]), s) for s in range(ns) if fill[s] < 4]
                    s = min(cand)[2]
                    load[s] += simd[i]
                    fill[s] += 1
                tot_sorted += int(load.max(axis=1).sum())
                continue
            pad = (-len(kbs)) % 4
            o = np.vstack([o, np.zeros((pad, 8), dtype=np.int64)]).reshape(-1, 4, 8).sum(axis=1)      # [super-step][wave]
            simd = o[:, :4] + o[:, 4:]
            tot_max_simd += int(simd.max(axis=1).sum())
            tot_max_wave += int(o.max(axis=1).sum())
            tot_steps += o.shape[0]
    print('super-steps %d: MFMA k-steps on the busiest SIMD now 8 per super-step (2 waves x 4 blocks); with a per-wave skip of empty '
          'sub-tiles %.2f (blocks as stored), %.2f (blocks dealt to super-steps by SIMD load); average over SIMDs %.2f'
          % (tot_steps, tot_max_simd / tot_steps, tot_sorted / tot_steps, n_sub / 4.0 / tot_steps))


if __name__ == '__main__':
    main()
