#!/usr/bin/env python3
"""MIM pretraining entry point -- same CLI and ini surface as the reference ``pretrain_mim.py``:

    python pretrain_mim.py <model_name> [-v verbose_iters] [-ct cp_minutes] [-dd data_dir]
    python -m torch.distributed.run --nproc-per-node N pretrain_mim.py <model_name> ...   (one process per GPU)

Reads ``configs/<model_name>.ini``, builds the model / optimiser / scheduler
(``utils.mim_vit.build_model``), streams HDF5 cutouts, runs ``run_iter`` per batch, evaluates the
validation loss every ``verbose_iters``, checkpoints ``models/<model_name>.pth.tar`` every
``cp_time`` minutes in the reference's format (batch_iters, losses, optimizer, lr_scheduler, model).
The linear-probe validation hook (``lp_class_data_file`` / ``lp_regress_data_file`` / ``lp_combine``) runs on rank 0; plots are
out of scope.
"""
import ast
import configparser
import os
import time
from collections import defaultdict

import numpy as np
import torch

from sky_embeddings_amd import distributed as sdist
from sky_embeddings_amd import ops
from utils.dataloaders import build_fits_dataloader, build_h5_dataloader
from utils.mim_vit import build_model
from utils.misc import parseArguments
from utils.pretrain_fns import linear_probe, run_iter


def _mean(vals):
    return float(torch.stack([torch.as_tensor(v, dtype=torch.float32).reshape(()).cpu() for v in vals]).mean()) if vals else float("nan")


def save_checkpoint(model_filename, cur_iter, losses, optimizer, lr_scheduler, model):
    torch.save({'batch_iters': cur_iter, 'losses': dict(losses), 'optimizer': optimizer.state_dict(),
                'lr_scheduler': lr_scheduler.state_dict(),
                'model': {k: v.detach().cpu() for k, v in model.module.state_dict().items()}}, model_filename)


def main(args):
    rank, world, local = sdist.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("pretrain_mim.py needs a GPU: the hot path is HIP-only (no CPU fallback)")
    device = torch.device('cuda', sdist.local_device_index(local))
    torch.cuda.set_device(device)
    if rank == 0:
        print(f'Using Torch version: {torch.__version__}')
        print(f'Using a {device} device, {world} process(es), one GPU each')
    cur_dir = os.path.dirname(os.path.abspath(__file__))
    config_dir, model_dir = os.path.join(cur_dir, 'configs/'), os.path.join(cur_dir, 'models/')
    data_dir = args.data_dir if args.data_dir is not None else os.path.join(cur_dir, 'data/')
    os.makedirs(model_dir, exist_ok=True)
    model_name = args.model_name
    config = configparser.ConfigParser()
    if not config.read(config_dir + model_name + '.ini'):
        raise FileNotFoundError(config_dir + model_name + '.ini')
    if rank == 0:
        print('\nCreating model: %s\n\nConfiguration:' % model_name)
        for key_head in config.keys():
            if key_head == 'DEFAULT':
                continue
            print('  %s' % key_head)
            for key in config[key_head].keys():
                print('    %s: %s' % (key, config[key_head][key]))
    model_filename = os.path.join(model_dir, model_name + '.pth.tar')
    torch.manual_seed(0)  # identical initial weights on every rank
    model, losses, cur_iter, optimizer, lr_scheduler = build_model(config, model_filename, device, build_optimizer=True)
    # from here on every rank draws its OWN random stream (MAE masking noise inside TrainStep, SimMIM mask generators in the
    # loader workers): seed = base + rank, so the global batch sees world x the mask diversity
    torch.manual_seed(1 + rank)
    optimizer.grad_scale = 1.0 / world

    if 'mim' in config['ARCHITECTURE']['model_type']:
        mask_ratio, max_mask_ratio = None, float(config['TRAINING']['max_mask_ratio'])
    else:
        mask_ratio, max_mask_ratio = float(config['TRAINING']['mask_ratio']), None
    from_tiles = 'train_data_file' not in config['DATA']      # the reference's shipped MIM configs: survey tiles (train_data_paths)
    num_workers = max(1, min(os.cpu_count() // max(world, 1), 12) - 1)
    common = dict(batch_size=int(config['TRAINING']['batch_size']), num_workers=num_workers,
                  patch_size=int(config['ARCHITECTURE']['patch_size']),
                  num_channels=int(config['ARCHITECTURE']['num_channels']), max_mask_ratio=max_mask_ratio,
                  img_size=int(config['ARCHITECTURE']['img_size']), num_patches=model.module.patch_embed.num_patches)
    sampler = None
    train_file = dataloader_train = tile_loader = None
    if from_tiles:
        # survey tiles (FITS): every item is one sky patch cut into cutouts_per_tile windows on the GPU
        # (utils/dataloaders.py:538-654); ranks take every world-th patch, the same number each
        from utils.misc import str2bool
        tile_loader = build_fits_dataloader(ast.literal_eval(config['DATA']['train_data_paths']), bands=ast.literal_eval(config['DATA']['bands']),
                                            min_bands=int(config['DATA']['min_bands']), batch_size=common['batch_size'],
                                            num_workers=num_workers, patch_size=common['patch_size'], max_mask_ratio=None,
                                            img_size=common['img_size'], cutouts_per_tile=int(config['DATA']['cutouts_per_tile']),
                                            use_calexp=str2bool(config['DATA'].get('use_calexp', 'True')), ra_dec=True,
                                            augment=False, shuffle=True, device=device)
        tiles = tile_loader.dataset.band_filenames
        tile_loader.dataset.band_filenames = tiles[rank:len(tiles) // world * world:world]
        if not tile_loader.dataset.band_filenames:
            raise SystemExit(f"no survey tiles with the requested bands under {config['DATA']['train_data_paths']}")
    else:
        train_file = os.path.join(data_dir, config['DATA']['train_data_file'])
        if world > 1:
            from sky_embeddings_amd.hdf5_lite import File
            with File(train_file) as f:
                n_train = len(f['cutouts'])
            sampler = sdist.DistributedIndexSampler(n_train, rank, world, shuffle=True, seed=1234)
        dataloader_train = build_h5_dataloader(train_file, shuffle=True, sampler=sampler, **common)
    # Training takes the fast path: the batched HDF5->HBM feeder (no per-item python; chunked files un-chunked once) feeding
    # the HIP-graph TrainStep (forward + staged backward, gradient all-reduce overlapped, fused AdamW, cosine LR).  The
    # per-item loader above still serves SimMIM mask generation, ragged final batches and validation.
    fast = os.environ.get("SKYEMB_FAST_TRAIN", "1") != "0"
    train_step = None
    if fast:
        from sky_embeddings_amd.feeder import CutoutFeeder
        from sky_embeddings_amd.train_step import TrainStep
        # SimMIM mode: the per-channel patch masks are drawn on the device inside the step (max_mask_ratio)
        train_step = TrainStep(model.module.engine, optimizer, lr_scheduler, common['batch_size'], mask_ratio=mask_ratio,
                               world_size=world, max_mask_ratio=max_mask_ratio)
    sharded = train_step is not None and getattr(train_step, "shard_optimizer", False)
    dataloader_val = build_h5_dataloader(os.path.join(data_dir, config['DATA']['val_data_file']), shuffle=True, **common)
    if rank == 0:
        if from_tiles:
            print('The training set consists of %i sky patches per process, %s cutouts each.' %
                  (len(tile_loader), config['DATA']['cutouts_per_tile']))
        else:
            print('The training set consists of %i cutouts.' % (len(dataloader_train.dataset)))

    lp_files = {key: os.path.join(data_dir, config['DATA'][key]) if key in config['DATA'] else None
                for key in ('lp_class_data_file', 'lp_regress_data_file')}
    lp_combine = config['DATA'].get('lp_combine', 'central')
    total_batch_iters = int(float(config['TRAINING']['total_batch_iters']))
    if rank == 0:
        print('Training the network with a batch size of %i per GPU ...' % (common['batch_size']))
        print('Progress will be displayed every %i batch iterations and the model will be saved every %i minutes.' %
              (args.verbose_iters, args.cp_time))
    losses_cp = defaultdict(list)
    cp_start_time = time.time()
    epoch = 0
    done = False
    while cur_iter < total_batch_iters and not done:
        if sampler is not None:
            sampler.set_epoch(epoch)
        epoch += 1
        if from_tiles:
            np.random.seed(1234 + epoch * world + rank)          # window corners: numpy's global generator, per rank and epoch

            def tile_batches():                                   # pretrain_mim.py:143-150 (get_train_samples, nested batches)
                for cut, msk, rds in tile_loader:
                    for b in range(cut.shape[1]):
                        yield cut[0, b], (msk[0, b] if msk.dim() == 6 else None), rds[0, b]
            loader = tile_batches()
        elif fast:
            loader = CutoutFeeder(train_file, common['batch_size'], common['img_size'], device, shuffle=True, seed=1234 + epoch,
                                  rank=rank, world_size=world, drop_last=world > 1)
        else:
            loader = dataloader_train
        epoch_start_iter = cur_iter
        for samples, masks, ra_decs in loader:
            if fast and samples.shape[0] == common['batch_size']:
                loss = train_step(samples, None, ra_decs)
                losses_cp['train_loss'].append(loss.detach().clone())
            else:
                samples = samples.to(device, non_blocking=True)
                if (fast or from_tiles) and max_mask_ratio is not None:
                    # (a ragged last batch of the feeder, or tile batches on the eager path) SimMIM masks from the device generator
                    eng_cfg = model.module.engine.cfg
                    masks = torch.empty_like(samples)
                    ops.simmim_mask_from_noise(torch.rand(samples.shape[0], eng_cfg.in_chans, eng_cfg.num_patches, device=device),
                                               torch.rand(samples.shape[0], device=device), max_mask_ratio, eng_cfg.grid,
                                               eng_cfg.patch_size, masks)
                # forward + backward, then the RCCL gradient all-reduce, then AdamW (run_iter's order)
                model.train(True)
                loss, _, _ = model(samples, ra_dec=ra_decs, mask_ratio=mask_ratio, mask=masks)
                loss.backward()
                sdist.allreduce_flat_gradients(model.module.engine.store.g, world)
                optimizer.step()
                optimizer.zero_grad(set_to_none=True)
                lr_scheduler.step()
                losses_cp['train_loss'].append(loss.detach())
            if cur_iter % args.verbose_iters == 0:
                for i, (vs, vm, vr) in enumerate(dataloader_val):
                    model, optimizer, lr_scheduler, losses_cp = run_iter(model, vs.to(device, non_blocking=True), vr, vm,
                                                                         mask_ratio, optimizer, lr_scheduler, losses_cp,
                                                                         mode='val')
                    if i >= 200:
                        break
                probing = rank == 0 and any(lp_files.values())
                if probing:     # the encoder is replicated: rank 0's probe is every rank's
                    linear_probe(model, losses_cp, device, dataloader_val, lp_files['lp_class_data_file'],
                                 lp_files['lp_regress_data_file'], combine=lp_combine)
                    model.train(True)       # the probe leaves the model in eval mode (utils/pretrain_fns.py:62)
                if world > 1 and any(lp_files.values()):
                    # the other ranks wait for rank 0's host-side scikit-learn fits HERE, on the host, not inside the next
                    # step's gradient all-reduce (whose watchdog would abort the job after ten minutes)
                    sdist.host_barrier()
                for k in list(losses_cp.keys()):
                    losses[k].append(_mean(losses_cp[k]))
                losses['batch_iters'].append(cur_iter)
                if rank == 0:
                    print('\nBatch Iterations: %i/%i ' % (cur_iter, total_batch_iters))
                    print('Losses:\n\tTraining Dataset\n\t\tTotal Loss: %0.3f' % (losses['train_loss'][-1]))
                    print('\tValidation Dataset\n\t\tTotal Loss: %0.3f' % (losses['val_loss'][-1]))
                    if probing:
                        print('Linear Probing Results:')
                        if lp_files['lp_class_data_file']:
                            print('\tClassification Accuracy:\n\t\tTraining: %0.3f, Validation: %0.3f' %
                                  (losses['train_lp_acc'][-1], losses['val_lp_acc'][-1]))
                        if lp_files['lp_regress_data_file']:
                            print('\tRegression R2\n\t\tTraining: %0.3f, Validation: %0.3f' %
                                  (losses['train_lp_r2'][-1], losses['val_lp_r2'][-1]))
                losses_cp = defaultdict(list)
            cur_iter += 1
            due = (time.time() - cp_start_time) >= args.cp_time * 60
            if sharded:
                # the optimiser is sharded over the ranks (TrainStep): a checkpoint starts with a collective gather of the fp32 state,
                # so the ranks take the wall-clock decision together -- rank 0's, looked at every 50 iterations
                due = cur_iter % 50 == 0 and sdist.agree(due)
            if due:
                if sharded:
                    train_step.gather_full_state()
                if rank == 0:
                    print('Saving network...')
                    save_checkpoint(model_filename, cur_iter, losses, optimizer, lr_scheduler, model)
                cp_start_time = time.time()
            if cur_iter > total_batch_iters:
                if sharded:
                    train_step.gather_full_state()
                if rank == 0:
                    print('Saving network...')
                    save_checkpoint(model_filename, cur_iter, losses, optimizer, lr_scheduler, model)
                done = True
                break
        if cur_iter == epoch_start_iter:
            # an epoch without a single batch (fewer cutouts per rank than one batch): the loop would never end
            raise RuntimeError('the training data yields no batch of %i samples per rank (world size %i)'
                               % (common['batch_size'], world))
    if os.environ.get('SKYEMB_SAVE_RANK_PARAMS'):
        # test hook: every rank's flat fp32 master weights, to check that the replicas stayed identical
        if sharded:
            train_step.gather_full_state()
        torch.cuda.synchronize(device)
        torch.save(model.module.engine.store.p.cpu(), os.path.join(os.environ['SKYEMB_SAVE_RANK_PARAMS'], f'rank{rank}.pt'))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(parseArguments().parse_args())
    print('\nTraining complete.')
