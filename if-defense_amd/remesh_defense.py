"""Reconstruct a mesh with the Occupancy Network and re-sample its surface - MI355X build of ONet/remesh_defense.py.

Same flags, defaults and .npz in/out as the reference CLI (ONet/remesh_defense.py:19-41,173-285; output
<dir>/ONet-Mesh/onet_remesh-<name>):

    python -m ifdefense_amd.remesh_defense --data_root=path/to/adv_data.npz

Additions: --seed (surface samples and the 300-point subset are counter-based draws; the reference is unseeded),
--weights.  Under torchrun the clouds of each file are sharded over the ranks and all-gathered (rank 0 writes).
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

from .opt_defense import check_supported, list_inputs, load_config, str2bool, validate_limits


def build_parser():
    parser = argparse.ArgumentParser(description='Extract meshes from occupancy process.')
    parser.add_argument('--config', type=str, default='configs/onet_mn40.yaml', help='Path to config file.')
    parser.add_argument('--sample_npoint', type=int, default=1024, help='Re-sample points number per mesh.')
    parser.add_argument('--padding_scale', type=float, default=0.9,
                        help='Used in pre-processing point clouds, padding in unit cube')
    parser.add_argument('--data_root', type=str, default='', help='Path to point cloud npz file.')
    parser.add_argument('--train', type=str2bool, default=False, help='whether defend training data')
    parser.add_argument('--sor', type=str2bool, default=True, help='whether use SOR before reconstruction')
    parser.add_argument('--sor_k', type=int, default=2, help='KNN in SOR')
    parser.add_argument('--sor_alpha', type=float, default=1.1, help='Threshold = mean + alpha * std')
    parser.add_argument('--seed', type=int, default=0, help='seed of the counter-based random draws')
    parser.add_argument('--weights', type=str, default='', help='checkpoint (.pth); default: cfg test.model_file')
    parser.add_argument('--precision', choices=('f32', 'bf16x6', 'bf16x3'), default='f32',
                        help="arithmetic of the decoder layers in the MISE grid evaluation (opt-in extension, ifd_mesh_params.precision; "
                             "default f32: the grid is bit-identical to the reference's MISE class on the same decoder values)")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    cfg = load_config(args.config, 'configs/default.yaml', "onet")
    check_supported(cfg, "onet")
    gen = cfg.get('generation', {}) or {}
    if gen.get('resolution_0', 32) != 32 or gen.get('upsampling_steps', 2) != 2 or gen.get('refinement_step', 0) != 0 or \
            gen.get('simplify_nfaces') is not None:
        raise SystemExit("unsupported generation config: only resolution_0 32 / upsampling_steps 2 without refinement or "
                         "simplification (configs/default.yaml:64-74) is built for MI355X")

    files = list_inputs(args.data_root, args.train)
    args.batch_size, args.iterations = 1, 0                       # (not flags of this CLI; validate_limits looks at them)
    validate_limits(args, cfg, files, "onet")

    import torch
    from . import DefenseArgs, OnetRestorer, defend_npz_test_data, remesh_point_cloud, weights, get_save_name
    from . import dist as D

    rank, world, local = D.init_from_env()
    r = OnetRestorer(weights.load_checkpoint(args.weights or cfg['test']['model_file'], "onet"),
                     device=torch.device('cuda', local), threshold=cfg['test']['threshold'])
    dargs = DefenseArgs(sample_npoint=args.sample_npoint, padding_scale=args.padding_scale, sor=args.sor, sor_k=args.sor_k,
                        sor_alpha=args.sor_alpha, threshold=cfg['test']['threshold'], input_npoint=cfg['data']['pointcloud_n'], precision=args.precision,
                        seed=args.seed)

    def defend(pc, normalize=True):
        out = D.defend_sharded(lambda shard, base, total: remesh_point_cloud(r, shard, dargs, base, return_device=True,
                                                                            normalize=normalize), pc)
        return out.cpu().numpy()

    def one_file_work(path):
        if args.train:                                            # remesh_defense.py:187-223 (train split not normalised)
            npz = np.load(path)
            tr = defend(npz['train_pc'][..., :3], normalize=False)
            te = defend(npz['test_pc'][..., :3])
            if rank == 0:
                save_path = get_save_name(path, "onet-mesh")
                np.savez(save_path, train_pc=tr.astype(np.float32), test_pc=te.astype(np.float32),
                         train_label=npz['train_label'], test_label=npz['test_label'])
                print('defense result saved to {}'.format(save_path))
        elif rank == 0:
            defend_npz_test_data(r, path, dargs, defend=defend, save_model="onet-mesh")
        else:
            defend(np.load(path)['test_pc'][..., :3])

    for path in files:
        err = None
        try:
            one_file_work(path)
        except Exception as e:                  # noqa: BLE001  (the ranks agree on a status and stop together)
            err = e
        bad = D.any_rank_failed(err is not None, torch.device('cuda', local))
        if err is not None:
            raise SystemExit("rank %d failed on %s: %s: %s" % (rank, path, type(err).__name__, err))
        if bad:
            raise SystemExit("rank %d stops: another rank failed on %s" % (rank, path))
    D.shutdown()
    return 0


if __name__ == '__main__':
    sys.exit(main())
