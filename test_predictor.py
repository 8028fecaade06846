#!/usr/bin/env python3
"""Evaluation of a trained predictor -- same CLI and ini surface as the reference ``test_predictor.py``:

    python test_predictor.py <model_name> [-dd data_dir]

Loads ``models/<model_name>[_best].pth.tar`` (``utils.vit.build_model``), predicts the validation file (``utils.eval_fns.ft_predict``),
keeps the objects whose minimum S/N over the first five channels exceeds 5 (``utils.misc.h5_snr``) and evaluates: redshift
regressions by the photo-z metrics of the whole set and per redshift / S/N bin, classifiers by their confusion matrix.  Where the
reference saves figures into ``figures/``, the numbers those figures show are saved as ``figures/<same name>.npz``
(``utils.plotting_fns``; rendering is out of scope).  The encoder runs in the HIP engine.
"""
import ast
import configparser
import os

import numpy as np
import torch

from utils.dataloaders import build_h5_dataloader
from utils.eval_fns import ft_predict
from utils.misc import h5_snr, parseArguments, str2bool
from utils.plotting_fns import evaluate_z, plot_conf_mat, plot_progress, plot_resid_hexbin
from utils.vit import build_model


def load_predictor(model_name, config_dir, model_dir, device):
    """(config, mae_config, model, losses, cur_iter, model_filename) of a trained predictor; the best checkpoint when there is one."""
    config = configparser.ConfigParser()
    if not config.read(config_dir + model_name + '.ini'):
        raise FileNotFoundError(config_dir + model_name + '.ini')
    print('\nCreating model: %s\n\nConfiguration:' % model_name)
    for key_head in config.keys():
        if key_head == 'DEFAULT':
            continue
        print('  %s' % key_head)
        for key in config[key_head].keys():
            print('    %s: %s' % (key, config[key_head][key]))
    model_filename = os.path.join(model_dir, model_name + '.pth.tar')
    if os.path.exists(model_filename.replace('.pth.tar', '_best.pth.tar')):
        model_filename = model_filename.replace('.pth.tar', '_best.pth.tar')
    mae_name = config['TRAINING']['pretained_mae']
    if mae_name == 'None':
        mae_filename, mae_config = 'None', config
    else:
        mae_config = configparser.ConfigParser()
        mae_config.read(config_dir + mae_name + '.ini')
        mae_filename = os.path.join(model_dir, mae_name + '.pth.tar')
    model, losses, cur_iter = build_model(config, mae_config, model_filename, mae_filename, device, build_optimizer=False)
    return config, mae_config, model, losses, cur_iter, model_filename


def validation_loader(config, mae_config, model, data_dir):
    return build_h5_dataloader(os.path.join(data_dir, config['DATA']['val_data_file']), batch_size=int(config['TRAINING']['batch_size']),
                               num_workers=max(1, min(os.cpu_count(), 12) - 1), label_keys=ast.literal_eval(config['DATA']['label_keys']),
                               img_size=int(config['ARCHITECTURE']['img_size']), patch_size=int(mae_config['ARCHITECTURE']['patch_size']),
                               num_channels=int(mae_config['ARCHITECTURE']['num_channels']), num_patches=model.module.patch_embed.num_patches,
                               shuffle=False)


def main(args):
    if not torch.cuda.is_available():
        raise SystemExit("test_predictor.py needs a GPU: the encoder is HIP-only (no CPU fallback)")
    device = torch.device('cuda')
    print(f'Using Torch version: {torch.__version__}')
    cur_dir = os.path.dirname(os.path.abspath(__file__))
    config_dir, model_dir, fig_dir = os.path.join(cur_dir, 'configs/'), os.path.join(cur_dir, 'models/'), os.path.join(cur_dir, 'figures/')
    data_dir = args.data_dir if args.data_dir is not None else os.path.join(cur_dir, 'data/')
    model_name = args.model_name
    config, mae_config, model, losses, _, model_filename = load_predictor(model_name, config_dir, model_dir, device)
    loss_fn = config['TRAINING']['loss_fn']
    plot_progress(losses, savename=os.path.join(fig_dir, f'{os.path.basename(model_filename).split(".")[0]}_progress.png'))
    dataloader_val = validation_loader(config, mae_config, model, data_dir)
    print('The validation set consists of %i cutouts.' % (len(dataloader_val.dataset)))
    tgt_labels, pred_labels = ft_predict(model, dataloader_val, device, use_label_errs=str2bool(config['TRAINING'].get('use_label_errs', 'False')))
    snr_vals = h5_snr(h5_path=os.path.join(data_dir, config['DATA']['val_data_file']), n_central_pix=8, batch_size=5000, num_samples=None)
    print(snr_vals.shape)
    snr = np.nanmin(snr_vals[:, :5], axis=1)          # minimum S/N of the (first) five channels
    snr_indices = snr > 5                             # only objects that are not super noisy
    print(len(np.where(snr_indices)[0]))
    if 'mse' in loss_fn.lower():
        plot_resid_hexbin([r'$Z$'], tgt_labels[snr_indices], pred_labels[snr_indices], y_lims=[1], gridsize=(80, 40), max_counts=5, n_std=4,
                          savename=os.path.join(fig_dir, f'{model_name}_predictions.png'))
        res = evaluate_z(pred_labels[snr_indices], tgt_labels[snr_indices], n_bins=8, z_range=(0.2, 1.6), threshold=0.1, snr=snr[snr_indices],
                         savename=os.path.join(fig_dir, f'{model_name}_redshift.png'))
        print('Bias: %0.4f  MAD: %0.4f  Outlier fraction: %0.4f' % (res['bias'], res['mad'], res['frac_out']))
    else:
        pred_class = np.argmax(pred_labels, 1)        # logits -> classes
        tgt_class = tgt_labels[:, 0]
        cm = plot_conf_mat(tgt_class[snr_indices], pred_class[snr_indices], ['galaxy', 'qso', 'star'],
                           savename=os.path.join(fig_dir, f'{model_name}_classes.png'))
        print('Confusion matrix (true class x predicted class):\n%s\nAccuracy: %0.3f' % (cm, np.trace(cm) / max(1, cm.sum())))


if __name__ == "__main__":
    main(parseArguments().parse_args())
    print('\nTesting complete.')
