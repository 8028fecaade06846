#!/usr/bin/env python3
"""Similarity search entry point -- same CLI flags as the reference ``similarity_search.py``
(-tgt_fn -tst_fn -tgt_i -aug -mp -ct -snr -bs -m -c -dc -np -ns -dd) and the same ``.npz`` output
(``results/<model>_<target>_simsearch_results_f.npz`` with test_ra_decs, test_scores, target_images,
target_features, test_images, test_features).

Differences: figures are not drawn (matplotlib/LaTeX plotting is out of scope);
the 64x target augmentation (``-aug True``, the reference's default) runs on the device
(sky_embeddings_amd.augment: torchvision's parameter draws, one HIP launch per batch).  ``--bank`` (extension) encodes the
test set ONCE into a resident embedding bank and runs the fused cosine top-k kernel over it
instead of re-scoring streamed batches (requires -mp True or -ct True, i.e. one vector per sample).
"""
import argparse
import ast
import configparser
import os

import numpy as np
import torch

from utils.dataloaders import build_h5_dataloader
from utils.eval_fns import build_embedding_bank, mae_latent
from utils.mim_vit import build_model as build_mim
from utils.misc import h5_snr, str2bool
from utils.similarity import determine_target_features, mae_simsearch
from utils.vit import build_model as build_vit


def parseArguments():
    parser = argparse.ArgumentParser('Similarity searching.', add_help=False)
    parser.add_argument("model_name", help="Name of model.", type=str)
    parser.add_argument("-tgt_fn", "--target_fn", type=str, default='HSC_dud_dwarf_galaxy_calexp_GIRYZ7610_64.h5')
    parser.add_argument("-tst_fn", "--test_fn", type=str, default='HSC_dud_unknown_calexp_GIRYZ7610_64.h5')
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
    parser.add_argument("-dd", "--data_dir", help="Data directory if different from sky_embeddings/data/", type=str,
                        default=None)
    parser.add_argument("--bank", action="store_true", help="encode once into a resident bank + fused top-k kernel")
    return parser


def main():
    args = parseArguments().parse_args()
    target_indices = ast.literal_eval(args.target_indices) if args.target_indices != 'None' else None
    max_pool, cls_token = str2bool(args.max_pool), str2bool(args.cls_token)
    snr_range = ast.literal_eval(args.snr_range)
    cur_dir = os.path.dirname(os.path.abspath(__file__))
    config_dir, model_dir = os.path.join(cur_dir, 'configs/'), os.path.join(cur_dir, 'models/')
    data_dir = args.data_dir if args.data_dir is not None else os.path.join(cur_dir, 'data/')
    results_dir = os.path.join(cur_dir, 'results/')
    os.makedirs(results_dir, exist_ok=True)
    if not torch.cuda.is_available():
        raise SystemExit("similarity_search.py needs a GPU: the hot path is HIP-only (no CPU fallback)")
    device = torch.device('cuda')
    print(f'Using Torch version: {torch.__version__}')
    config = configparser.ConfigParser()
    config.read(config_dir + args.model_name + '.ini')
    model_filename = os.path.join(model_dir, args.model_name + '.pth.tar')
    if 'pretained_mae' in config['TRAINING']:
        mae_name = config['TRAINING']['pretained_mae']
        if mae_name == 'None':
            mae_filename, mae_config = 'None', config
        else:
            mae_config = configparser.ConfigParser()
            mae_config.read(config_dir + mae_name + '.ini')
            mae_filename = os.path.join(model_dir, mae_name + '.pth.tar')
        model, losses, cur_iter = build_vit(config, mae_config, model_filename, mae_filename, device)
    else:
        mae_config = config
        model, losses, cur_iter = build_mim(config, model_filename, device, build_optimizer=False)

    print('Estimating S/N for test dataset images...')
    test_snr = h5_snr(os.path.join(data_dir, args.test_fn), n_central_pix=8, batch_size=5000)
    test_snr = np.nanmin(test_snr[:, :5], axis=1)
    test_indices = np.where((test_snr > snr_range[0]) & (test_snr < snr_range[1]))[0]
    common = dict(batch_size=args.batch_size, num_workers=min(os.cpu_count(), 12),
                  img_size=int(config['ARCHITECTURE']['img_size']), num_patches=model.module.patch_embed.num_patches,
                  patch_size=int(mae_config['ARCHITECTURE']['patch_size']),
                  num_channels=int(mae_config['ARCHITECTURE']['num_channels']), max_mask_ratio=None, shuffle=False)
    target_dataloader = build_h5_dataloader(os.path.join(data_dir, args.target_fn), indices=target_indices, **common)
    test_dataloader = build_h5_dataloader(os.path.join(data_dir, args.test_fn), indices=test_indices, **common)
    target_latent, target_images = mae_latent(model, target_dataloader, device, return_images=True,
                                              apply_augmentations=str2bool(args.augment_targets), num_augmentations=64,
                                              remove_cls=False)
    if args.bank:
        from sky_embeddings_amd import search
        assert args.metric == 'cosine' and (max_pool or cls_token), "--bank scores one vector per sample"
        mod = model.module
        tl = target_latent.to(device)
        tl = tl[:, :1] if cls_token else tl[:, mod.num_extra_tokens:].max(dim=1, keepdim=True).values
        bank = build_embedding_bank(model, test_dataloader, device, pool='cls' if cls_token else 'max')
        first = bank[:args.batch_size]   # the reference standardises with the first batch (utils/similarity.py:98-100)
        mean_feats, std_feats = first.mean(dim=0), first.std(dim=0, unbiased=True)
        tl = (tl - mean_feats) / (std_feats + 1e-8)
        search.standardise_(bank, mean_feats, std_feats)
        avg, w = determine_target_features(tl)
        scores, idx = search.cosine_topk(avg.reshape(1, -1), bank, min(args.n_save, bank.shape[0]), weights=w)
        test_scores, order = scores[0], idx[0].cpu().numpy()
        ds = test_dataloader.dataset
        items = [ds[int(j)] for j in order if j >= 0]
        test_images = torch.stack([it[0] for it in items])
        test_ra_decs = torch.stack([it[2] for it in items])
        test_latent, _, _ = model.module.forward_features(test_images.to(device), reshape_out=False)
    else:
        test_images, test_latent, test_ra_decs, test_scores = mae_simsearch(
            model, target_latent, test_dataloader, device, metric=args.metric, combine=args.combine, use_weights=True,
            max_pool=max_pool, cls_token=cls_token, nested_batches=False, n_save=args.n_save)
    out = os.path.join(results_dir, f'{args.model_name}_{args.target_fn[:-3]}_simsearch_results_f.npz')
    np.savez(out, test_ra_decs=test_ra_decs.cpu().numpy(), test_scores=test_scores.cpu().numpy(),
             target_images=target_images.cpu().numpy(), target_features=target_latent.cpu().numpy(),
             test_images=test_images.cpu().numpy(), test_features=test_latent.cpu().numpy())
    print('saved', out)


if __name__ == "__main__":
    main()
