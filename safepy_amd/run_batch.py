"""Batch driver: the reference's command-line entry (`python safe.py <attribute_file>`,
safepy/safe.py:1309-1361) re-expressed as the attribute-sharded multi-GPU launcher.

    python -m safepy_amd.run_batch ATTRIBUTE_FILE --network NET.scatter|NET.gpickle [options]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
           -m safepy_amd.run_batch ATTRIBUTE_FILE --network NET [options]

The reference reads the attribute file once, splits the columns with `np.array_split` over
`cpu_count()` worker processes, runs load_network / define_neighborhoods / load_attributes /
compute_pvalues(num_permutations=1000) in each and pickles `np.concatenate(all NES blocks,
axis=1)` to `<attribute_file>_safe_nes.p`.  Here: one process per GPU (torch.distributed, RCCL),
the same `array_split` of the columns over the ranks, membership and permutation stream
replicated, whole-matrix dispatch statistics (sharding.sharded_compute_pvalues), one final
all-gather; rank 0 writes the same pickle.  Without a launcher (WORLD_SIZE unset) it runs on one
GPU.  The default Costanzo network of the reference's safe-data repository is not bundled, so
--network is required."""
import argparse
import logging
import os
import pickle
import time

import numpy as np


def main(argv=None):
    ap = argparse.ArgumentParser(description='Run SAFE on the columns of an attribute file, sharded across the GPUs of a node')
    ap.add_argument('path_to_attribute_file', type=str, help='label-to-attribute annotations (.txt / .gz)')
    ap.add_argument('--network', required=True, help='.scatter or .gpickle network file')
    ap.add_argument('--node-key', default=None, help="node attribute matched against the attribute file's labels "
                                                    "(default: 'key' for .scatter, 'label_orf' otherwise)")
    ap.add_argument('--metric', default=None, choices=['euclidean', 'shortpath', 'shortpath_weighted_layout'])
    ap.add_argument('--radius', type=float, default=None)
    ap.add_argument('--how', default='auto')
    ap.add_argument('--permutations', type=int, default=1000)
    ap.add_argument('--score', default='sum', choices=['sum', 'z-score'])
    ap.add_argument('--seed', type=int, default=None)
    ap.add_argument('--multiple-testing', action='store_true', help='Benjamini-Hochberg per row across all attributes (safe.py:536-542)')
    ap.add_argument('--output', default=None, help='default: <attribute_file>_safe_nes.p (safe.py:1357)')
    args = ap.parse_args(argv)
    start = time.time()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # RCCL brings its own HIP streams: with the runtime's default of four hardware queues the context's kernel / table streams
    # end up sharing queues with them (measured: +0.8 ms on a 4.2 ms permutation step).  More queues, and the context first.
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    import safepy_amd
    from safepy_amd import sharding, backend
    # ranks of one node share the host: sleeping host waits when the share
    # is under three cores; the permutation stream itself is drawn once per node (sharding.ensure_shared_stream)
    backend.configure_host_for_ranks(int(os.environ.get('LOCAL_WORLD_SIZE', str(world))))
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        backend.Context.default(local_rank)              # (its streams exist before the process group's)
        dist.init_process_group('nccl')
    backend.pin_threads_to_device_numa(local_rank)       # this rank's host threads next to its GPU
    sf = safepy_amd.SAFE(verbose=(rank == 0), device=local_rank)
    if args.seed is not None:
        sf.random_seed = args.seed
    if args.radius is not None:
        sf.neighborhood_radius = args.radius
    is_scatter = args.network.endswith('.scatter')
    sf.load_network(network_file=args.network, node_key_attribute=args.node_key or ('key' if is_scatter else 'label_orf'),
                    pseudo_network='arrays')
    kw = {}
    if args.metric or is_scatter:
        kw['node_distance_metric'] = args.metric or 'euclidean'          # a .scatter network has no edges
    sf.define_neighborhoods(**kw)
    sf.load_attributes(attribute_file=args.path_to_attribute_file)      # every rank parses the same file: no broadcast
    m_total = sf.node2attribute.shape[1]

    if world == 1:
        sf.compute_pvalues(how=args.how, num_permutations=args.permutations, neighborhood_score_type=args.score,
                           multiple_testing=args.multiple_testing)
        all_nes = sf.nes
    else:
        c0, c1 = sharding.column_shards(m_total, world)[rank]
        if rank == 0:
            logging.info('Running SAFE on %d shards of about %d attributes...' % (world, -(-m_total // world)))
        out = sharding.sharded_compute_pvalues(
            sf._ctx(), sf._device_neighborhoods(), np.ascontiguousarray(sf.node2attribute[:, c0:c1]), m_total,
            enrichment_type=args.how, num_permutations=args.permutations, random_seed=sf.random_seed,
            neighborhood_score_type=args.score, attribute_sign=sf.attribute_sign,
            enrichment_threshold=sf.enrichment_threshold, gather=('nes',), multiple_testing=args.multiple_testing)
        all_nes = out['full_nes']

    if rank == 0:
        output_file = args.output or format('%s_safe_nes.p' % args.path_to_attribute_file)
        logging.info('Saving the results...')
        with open(output_file, 'wb') as handle:
            pickle.dump(all_nes, handle)
        print('%s: NES %d x %d, %.2f s' % (output_file, all_nes.shape[0], all_nes.shape[1], time.time() - start))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
