#!/usr/bin/env python3
"""Scores of families of trained predictors against their training-set size -- the reference's ``compare_predictors.py``:

    python compare_predictors.py <any name> [-dd data_dir]

For every model of the reference's hard-coded families (fully supervised / fine-tuned / attentive probe / wide / large, 2^7 .. 2^14
training samples) whose ini and checkpoint exist under ``configs/`` and ``models/``: the validation predictions
(``utils.eval_fns.ft_predict``) and, for redshift regressions, bias / MAD / mean squared error (``photoz_prediction_metrics``), for
classifiers the accuracy.  The reference draws ``figures/numsamples_redshift.png`` / ``numsamples_class.png``; here the score table
those figures show is saved as ``figures/numsamples_redshift.npz`` / ``numsamples_class.npz`` (missing models are NaN).
"""
import os

import numpy as np
import torch

from test_predictor import load_predictor, validation_loader
from utils.eval_fns import ft_predict
from utils.misc import parseArguments, str2bool
from utils.plotting_fns import photoz_prediction_metrics

categories = ['Fully Supervised', 'Fine-tuning', 'Attentive Probing', 'Fine-tuning (Wide)', 'Fine-tuning (Wide+Large)']
num_samples = (2 ** np.arange(7, 15)).astype(int)
_sizes = ['012k', '025k', '05k', '1k', '2k', '4k', '8k', '16k']
model_names = [['cls_fs_%s' % s for s in _sizes], ['cls_ft_%s' % s for s in _sizes], ['cls_ap_%s' % s for s in _sizes],
               ['cls_ft_%s_wide' % s for s in _sizes], ['cls_ft_%s_large' % s for s in _sizes]]


def main(args):
    if not torch.cuda.is_available():
        raise SystemExit("compare_predictors.py needs a GPU: the encoder is HIP-only (no CPU fallback)")
    device = torch.device('cuda')
    cur_dir = os.path.dirname(os.path.abspath(__file__))
    config_dir, model_dir, fig_dir = os.path.join(cur_dir, 'configs/'), os.path.join(cur_dir, 'models/'), os.path.join(cur_dir, 'figures/')
    data_dir = args.data_dir if args.data_dir is not None else os.path.join(cur_dir, 'data/')
    scores = np.full((len(categories), 3, len(num_samples)), np.nan)
    loss_fn = None
    for i in range(len(categories)):
        for j, model_name in enumerate(model_names[i]):
            if not (os.path.exists(config_dir + model_name + '.ini') and
                    any(os.path.exists(os.path.join(model_dir, model_name + sfx)) for sfx in ('.pth.tar', '_best.pth.tar'))):
                print('Skipping %s (no configuration or checkpoint)' % model_name)
                continue
            config, mae_config, model, _, _, _ = load_predictor(model_name, config_dir, model_dir, device)
            loss_fn = config['TRAINING']['loss_fn']
            dataloader_val = validation_loader(config, mae_config, model, data_dir)
            print('The validation set consists of %i cutouts.' % (len(dataloader_val.dataset)))
            tgt_labels, pred_labels = ft_predict(model, dataloader_val, device, use_label_errs=str2bool(config['TRAINING'].get('use_label_errs', 'False')))
            if 'mse' in loss_fn.lower():
                _, bias, mad, _ = photoz_prediction_metrics(pred_labels, tgt_labels, threshold=0.15)
                scores[i, 0, j], scores[i, 1, j], scores[i, 2, j] = bias, mad, np.mean((tgt_labels - pred_labels) ** 2)
            else:
                scores[i, 0, j] = np.mean(np.argmax(pred_labels, 1) == tgt_labels[:, 0])
    os.makedirs(fig_dir, exist_ok=True)
    name = 'numsamples_redshift' if (loss_fn is not None and 'mse' in loss_fn.lower()) else 'numsamples_class'
    np.savez(os.path.join(fig_dir, name + '.npz'), num_samples=num_samples, scores=scores, categories=np.array(categories),
             model_names=np.array(model_names))
    print('Scores [family, metric, training-set size]:\n%s' % np.array2string(scores, precision=4))


if __name__ == "__main__":
    main(parseArguments().parse_args())
    print('\nTesting complete.')
