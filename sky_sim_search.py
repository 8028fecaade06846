#!/usr/bin/env python3
"""Similarity search over SURVEY TILES -- same CLI flags as the reference ``sky_sim_search.py``
(-tgt_fn -tst_dirs -tgt_i -aug -mp -ct -snr -bs -m -c -dc -np -ns -dd) and the same ``.npz`` output
(``results/<model>_<target>_simsearch_results.npz`` with test_ra_decs, test_scores, target_images, target_features,
test_images, test_features).

Where ``similarity_search.py`` scores the cutouts of an HDF5 file, this entry point streams OVERLAPPING cutouts of every
FITS tile under ``--test_dirs`` (``build_fits_dataloader(..., use_overlap=True, overlap=0.4)``, sky_sim_search.py:137-150):
a tile's band files go to HBM as they are, all of its windows are cut in one launch (``skyemb_tile_cutouts``), encoded by
the HIP encoder and scored against the targets batch by batch (``mae_simsearch(nested_batches=True)``), the running best
``n_save`` kept on the device.  Figures are not drawn (plotting is out of scope); ``-snr`` is accepted
and unused, as in the reference (sky_sim_search.py parses it and never applies it to tiles).
"""
import argparse
import ast
import configparser
import os

import numpy as np
import torch

from utils.dataloaders import build_fits_dataloader, build_h5_dataloader
from utils.eval_fns import mae_latent
from utils.mim_vit import build_model as build_mim
from utils.misc import str2bool
from utils.similarity import mae_simsearch
from utils.vit import build_model as build_vit


def parseArguments():
    parser = argparse.ArgumentParser('Similarity searching.', add_help=False)
    parser.add_argument("model_name", help="Name of model.", type=str)
    parser.add_argument("-tgt_fn", "--target_fn", type=str, default='HSC_dud_dwarf_galaxy_calexp_GIRYZ7610_64.h5')
    parser.add_argument("-tst_dirs", "--test_dirs", type=str, nargs='+', default=['/project/rrg-kyi/astro/hsc/pdr3_dud/'])
    parser.add_argument("-tgt_i", "--target_indices", default='[1,2]')
    parser.add_argument("-aug", "--augment_targets", type=str, default='True')
    parser.add_argument("-mp", "--max_pool", type=str, default='True')
    parser.add_argument("-ct", "--cls_token", type=str, default='False')
    parser.add_argument("-snr", "--snr_range", default='[2,7]')
    parser.add_argument("-bs", "--batch_size", type=int, default=64)
    parser.add_argument("-m", "--metric", type=str, default='cosine')
    parser.add_argument("-c", "--combine", type=str, default='min')
    parser.add_argument("-dc", "--display_channel", type=int, default=2)
    parser.add_argument("-np", "--n_plot", type=int, default=36)
    parser.add_argument("-ns", "--n_save", type=int, default=300)
    parser.add_argument("-dd", "--data_dir", help="Data directory if different from sky_embeddings/data/", type=str, default=None)
    return parser


def main():
    args = parseArguments().parse_args()
    target_indices = ast.literal_eval(args.target_indices) if args.target_indices != 'None' else None
    max_pool, cls_token = str2bool(args.max_pool), str2bool(args.cls_token)
    cur_dir = os.path.dirname(os.path.abspath(__file__))
    config_dir, model_dir = os.path.join(cur_dir, 'configs/'), os.path.join(cur_dir, 'models/')
    data_dir = args.data_dir if args.data_dir is not None else os.path.join(cur_dir, 'data/')
    results_dir = os.path.join(cur_dir, 'results/')
    os.makedirs(results_dir, exist_ok=True)
    if not torch.cuda.is_available():
        raise SystemExit("sky_sim_search.py needs a GPU: the hot path is HIP-only (no CPU fallback)")
    device = torch.device('cuda')
    print(f'Using Torch version: {torch.__version__}')
    config = configparser.ConfigParser()
    if not config.read(config_dir + args.model_name + '.ini'):
        raise FileNotFoundError(config_dir + args.model_name + '.ini')
    model_filename = os.path.join(model_dir, args.model_name + '.pth.tar')
    if 'pretained_mae' in config['TRAINING']:
        mae_name = config['TRAINING']['pretained_mae']
        if mae_name == 'None':
            mae_filename, mae_config = 'None', config
        else:
            mae_config = configparser.ConfigParser()
            mae_config.read(config_dir + mae_name + '.ini')
            mae_filename = os.path.join(model_dir, mae_name + '.pth.tar')
        model, losses, cur_iter = build_vit(config, mae_config, model_filename, mae_filename, device, build_optimizer=False)
    else:
        mae_config = config
        model, losses, cur_iter = build_mim(config, model_filename, device, build_optimizer=False)

    target_dataloader = build_h5_dataloader(os.path.join(data_dir, args.target_fn), batch_size=args.batch_size,
                                            num_workers=min(os.cpu_count(), 12), img_size=int(config['ARCHITECTURE']['img_size']),
                                            num_patches=model.module.patch_embed.num_patches,
                                            patch_size=int(mae_config['ARCHITECTURE']['patch_size']),
                                            num_channels=int(mae_config['ARCHITECTURE']['num_channels']), max_mask_ratio=None,
                                            shuffle=False, indices=target_indices)
    test_dataloader = build_fits_dataloader(args.test_dirs, bands=ast.literal_eval(config['DATA']['bands']), min_bands=int(config['DATA']['min_bands']),
                                            batch_size=args.batch_size, num_workers=2, patch_size=int(config['ARCHITECTURE']['patch_size']),
                                            max_mask_ratio=None, img_size=int(config['ARCHITECTURE']['img_size']),
                                            cutouts_per_tile=int(config['DATA']['cutouts_per_tile']),
                                            use_calexp=str2bool(config['DATA']['use_calexp']), ra_dec=True, augment=False, shuffle=False,
                                            use_overlap=True, overlap=0.4, device=device)
    if len(test_dataloader) == 0:
        raise SystemExit(f"no survey tiles with the requested bands under {args.test_dirs}")
    print('Searching %i sky patch(es) in overlapping %sx%s cutouts...' % (len(test_dataloader), config['ARCHITECTURE']['img_size'],
                                                                          config['ARCHITECTURE']['img_size']))
    target_latent, target_images = mae_latent(model, target_dataloader, device, return_images=True,
                                              apply_augmentations=str2bool(args.augment_targets), num_augmentations=64, remove_cls=False)
    test_images, test_latent, test_ra_decs, test_scores = mae_simsearch(
        model, target_latent, test_dataloader, device, metric=args.metric, combine=args.combine, use_weights=True,
        max_pool=max_pool, cls_token=cls_token, nested_batches=True, n_save=args.n_save)
    out = os.path.join(results_dir, f'{args.model_name}_{args.target_fn[:-3]}_simsearch_results.npz')
    np.savez(out, test_ra_decs=test_ra_decs.cpu().numpy(), test_scores=test_scores.cpu().numpy(),
             target_images=target_images.cpu().numpy(), target_features=target_latent.cpu().numpy(),
             test_images=test_images.cpu().numpy(), test_features=test_latent.cpu().numpy())
    print('saved', out)


if __name__ == "__main__":
    main()
