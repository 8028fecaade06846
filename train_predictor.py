#!/usr/bin/env python3
"""Downstream predictor training -- same CLI and ini surface as the reference ``train_predictor.py``:

    python train_predictor.py <model_name> [-v verbose_iters] [-ct cp_minutes] [-dd data_dir]

Reads ``configs/<model_name>.ini`` (+ the pre-trained model's ini named by ``[TRAINING] pretained_mae``), builds the ViT
predictor on the pre-trained encoder with its optimiser (``utils.vit.build_model``: fine-tuning with layer-wise lr decay,
linear / attentive probe, or fully supervised), trains with ``utils.predictor_training_fns.run_iter`` on labelled HDF5 cutouts,
evaluates the validation set every ``verbose_iters``, keeps ``models/<name>_best.pth.tar`` (lowest validation loss; training
stops after 50 evaluations without improvement) and checkpoints ``models/<name>.pth.tar`` every ``cp_time`` minutes, in the
reference's format.  The encoder runs forward and backward in the HIP engine; progress plots are out of scope.
"""
import ast
import configparser
import os
import time
from collections import defaultdict

import numpy as np
import torch

from utils.dataloaders import build_h5_dataloader
from utils.misc import parseArguments, select_training_indices, str2bool
from utils.predictor_training_fns import run_iter
from utils.vit import build_model


def save_checkpoint(filename, cur_iter, losses, optimizer, lr_scheduler, model):
    torch.save({'batch_iters': cur_iter, 'losses': dict(losses), 'optimizer': optimizer.state_dict(), 'lr_scheduler': lr_scheduler.state_dict(),
                'model': {k: v.detach().cpu() for k, v in model.module.state_dict().items()}}, filename)


def split_labels(sample_labels, use_label_errs):
    """train_predictor.py:145-151: with label errors the second half of the label columns are the uncertainties."""
    if not use_label_errs:
        return sample_labels, None
    n = sample_labels.size(1) // 2
    return sample_labels[:, :n], sample_labels[:, n:]


def main(args):
    if not torch.cuda.is_available():
        raise SystemExit("train_predictor.py needs a GPU: the encoder is HIP-only (no CPU fallback)")
    device = torch.device('cuda')
    print(f'Using Torch version: {torch.__version__}')
    cur_dir = os.path.dirname(os.path.abspath(__file__))
    config_dir, model_dir = os.path.join(cur_dir, 'configs/'), os.path.join(cur_dir, 'models/')
    data_dir = args.data_dir if args.data_dir is not None else os.path.join(cur_dir, 'data/')
    os.makedirs(model_dir, exist_ok=True)
    model_name = args.model_name
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
    best_filename = model_filename.replace('.pth.tar', '_best.pth.tar')
    mae_name = config['TRAINING']['pretained_mae']
    if mae_name == 'None':
        mae_filename, mae_config = 'None', config
    else:
        mae_config = configparser.ConfigParser()
        mae_config.read(config_dir + mae_name + '.ini')
        mae_filename = os.path.join(model_dir, mae_name + '.pth.tar')
    model, losses, cur_iter, optimizer, lr_scheduler = build_model(config, mae_config, best_filename if os.path.exists(best_filename) else model_filename,
                                                                   mae_filename, device, build_optimizer=True)
    tr = config['TRAINING']
    loss_fn = tr['loss_fn']
    use_label_errs = str2bool(tr.get('use_label_errs', 'False'))
    num_workers = max(1, min(os.cpu_count(), 12) - 1)
    num_train = int(tr.get('num_train', '-1'))
    train_file = os.path.join(data_dir, config['DATA']['train_data_file'])
    if num_train > -1:
        train_indices = select_training_indices(train_file, num_train, balanced=False) if 'crossentropy' in loss_fn.lower() else range(num_train)
    else:
        train_indices = None
    common = dict(batch_size=int(tr['batch_size']), num_workers=num_workers, label_keys=ast.literal_eval(config['DATA']['label_keys']),
                  img_size=int(config['ARCHITECTURE']['img_size']), patch_size=int(mae_config['ARCHITECTURE']['patch_size']),
                  num_channels=int(mae_config['ARCHITECTURE']['num_channels']), num_patches=model.module.patch_embed.num_patches, shuffle=True)
    dataloader_train = build_h5_dataloader(train_file, augment=str2bool(tr.get('augment', 'False')), brightness=float(tr.get('brightness', '0.8')),
                                           noise=float(tr.get('noise', '0.01')), nan_channels=int(tr.get('nan_channels', '2')),
                                           indices=train_indices, **common)
    dataloader_val = build_h5_dataloader(os.path.join(data_dir, config['DATA']['val_data_file']), **common)
    print('The training set consists of %i cutouts.' % (len(dataloader_train.dataset)))
    total_batch_iters = int(float(tr['total_batch_iters']))
    print('Training the network with a batch size of %i per GPU ...' % (dataloader_train.batch_size))
    print('Progress will be displayed every %i batch iterations and the model will be saved every %i minutes.' % (args.verbose_iters, args.cp_time))
    metric = 'mae' if 'mse' in loss_fn.lower() else 'acc'
    best_val_loss = np.min(losses['val_loss']) if 'val_loss' in losses and len(losses['val_loss']) else np.inf
    did_not_improve_count = 0
    losses_cp = defaultdict(list)
    cp_start_time = time.time()
    done = False
    while cur_iter < total_batch_iters and did_not_improve_count < 50 and not done:
        for input_samples, sample_masks, ra_decs, sample_labels in dataloader_train:
            labels, label_errs = split_labels(sample_labels.to(device, non_blocking=True), use_label_errs)
            model, optimizer, lr_scheduler, losses_cp = run_iter(model, input_samples.to(device, non_blocking=True), None, ra_decs.to(device), labels,
                                                                 optimizer, lr_scheduler, losses_cp, loss_fn, label_uncertainties=label_errs, mode='train')
            if cur_iter % args.verbose_iters == 0:
                for vs, vm, vr, vl in dataloader_val:
                    labels, label_errs = split_labels(vl.to(device, non_blocking=True), use_label_errs)
                    model, optimizer, lr_scheduler, losses_cp = run_iter(model, vs.to(device, non_blocking=True), None, vr.to(device), labels, optimizer,
                                                                         lr_scheduler, losses_cp, loss_fn, label_uncertainties=label_errs, mode='val')
                for k in losses_cp.keys():
                    losses[k].append(float(np.mean(np.array(losses_cp[k]), axis=0)))
                losses['batch_iters'].append(cur_iter)
                print('\nBatch Iterations: %i/%i ' % (cur_iter, total_batch_iters))
                for tag, name in (('train', 'Training'), ('val', 'Validation')):
                    print('\t%s Dataset\n\t\tTotal Loss: %0.3e' % (name, losses[tag + '_loss'][-1]))
                    print(('\t\tMAE: %0.3e' if metric == 'mae' else '\t\tAccuracy: %0.3f') % (losses[f'{tag}_{metric}'][-1]))
                losses_cp = defaultdict(list)
                if losses['val_loss'][-1] < best_val_loss:
                    best_val_loss = losses['val_loss'][-1]
                    print('Saving network...')
                    save_checkpoint(best_filename, cur_iter, losses, optimizer, lr_scheduler, model)
                    did_not_improve_count = 0
                else:
                    did_not_improve_count += 1
            cur_iter += 1
            if (time.time() - cp_start_time) >= args.cp_time * 60:
                print('Saving network...')
                save_checkpoint(model_filename, cur_iter, losses, optimizer, lr_scheduler, model)
                cp_start_time = time.time()
            if cur_iter > total_batch_iters:
                print('Saving network...')
                save_checkpoint(model_filename, cur_iter, losses, optimizer, lr_scheduler, model)
                done = True
                break


if __name__ == "__main__":
    main(parseArguments().parse_args())
    print('\nTraining complete.')
