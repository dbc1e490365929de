"""Training loop with the reference's contract (Train/trainer1.py:14-255): loss_function,
KLAnnealer, save_checkpoint, run_epoch, train_model -- same arguments, history keys, log
line, CSV files and checkpoint dict -- on the HIP loss kernels (K6/K7).

Differences that do not change results: masks are built on the device; the three per-step
scalars cross to the host in ONE transfer instead of three `.item()` syncs; `np.float`
(removed from NumPy) is not used."""
import os
from time import time

import pandas as pd
import torch
import torch.distributed as dist

from .. import engine
from ..Model.forward_propagation1 import forward_propagation, prefetch


def KLAnnealer(epoch, KLA_ini_beta, KLA_inc_beta, KLA_beg_epoch):
    return KLA_ini_beta + KLA_inc_beta * ((epoch + 1) - KLA_beg_epoch)


def loss_function(beta, preds_prop, preds_mol, ys_cond, ys_mol, mu, log_var, use_cond2dec, pad_id):
    """RCE = CE(sum, ignore pad); KLD over every latent element; loss = RCE + beta*KLD
    (+ MSE_sum(prop) with use_cond2dec).  Returns (loss, RCE_mol, RCE_prop, KLD)."""
    RCE_mol = engine.CrossEntropySumFn.apply(preds_mol, ys_mol, pad_id)
    KLD = engine.KldFn.apply(mu, log_var)
    if use_cond2dec:
        diff = preds_prop - ys_cond          # [B, n_c, 1]: a few hundred floats, host-side glue
        RCE_prop = (diff * diff).sum()
        loss = RCE_mol + RCE_prop + beta * KLD
    else:
        RCE_prop = torch.zeros(1)
        loss = RCE_mol + beta * KLD
    return loss, RCE_mol, RCE_prop, KLD


def save_checkpoint(args, model, optimizer, save_path):
    names = ('N', 'd_model', 'd_ff', 'H', 'latent_dim', 'dropout', 'use_cond2dec', 'use_cond2lat',
             'variational')
    hyper = {'nconds': len(args.property_list)}
    for n in names:
        hyper[n] = getattr(args, n)
    torch.save({'model_state_dict': model.state_dict(), 'opt_state_dict': optimizer.state_dict(),
                'model_params': hyper}, save_path)


def warmup_lr(step, d_model, warm):
    """lr written after step k and used by step k+1 (reference trainer1.py:117-127)."""
    return float(d_model) ** -0.5 * min(float(step) ** -0.5, float(step) * float(warm) ** -1.5)


def run_epoch(args, model, optimizer, dataloader, current_step, beta, LOG, train):
    n_samples = 0
    every = max(1, int(getattr(args, 'print_every', 1)))
    history = {'RCE': [], 'KLD': [], 'LOSS': [], 'BETA': [], 'LR': []}
    model_cost_time = update_cost_time = 0
    cost_time = -time()
    nprop = len(args.property_list)
    n_batches = len(dataloader)
    pending = []                 # steps whose three loss scalars are on their way to the host

    def resolve():
        while pending:
            i_, n_, beta_, lr_, vals, ev = pending.pop(0)
            if ev is not None:
                ev.synchronize()
            rce, kld, tot = vals.tolist()
            history['RCE'].append(rce / n_)
            history['KLD'].append(kld / n_)
            history['LOSS'].append(tot / n_)
            history['BETA'].append(beta_)
            history['LR'].append(lr_)
            if (i_ + 1) % every == 0:
                LOG.info(f'{i_+1}/{n_batches:<10}\tRCE: {history["RCE"][-1]:.5f}\t'
                         f'KLD: {history["KLD"][-1]:.5f}\tLOSS: {history["LOSS"][-1]:.5f}\t'
                         f'TIME(s): {time() + cost_time:.1f}\tMODELTIME(s): {model_cost_time:.1f}\t'
                         f'UPDATETIME(s): {update_cost_time:.1f}')

    batches = iter(dataloader)
    upcoming = next(batches, None)
    i = -1
    while upcoming is not None:
        batch, upcoming, i = upcoming, next(batches, None), i + 1
        current_step += 1
        n_onebatch = batch['src'].size(0)
        model_cost_time -= time()
        # skip_ignored: the decoder rows of padded targets never reach this loss (ignore_index below) -- the model does
        # not compute them (engine.decoder_trunk_fwd; GCT_COMPACT_FWD=0 switches the shortcut off)
        preds_prop, preds_mol, mu, log_var, _ = forward_propagation[args.model_type](
            model, batch, args.pad_id, args.use_cond2dec, skip_ignored=True)[:5]
        if upcoming is not None:
            # the NEXT batch's masks and row maps are queued behind this forward: their read-back lands during this
            # step's backward, so the next forward starts without a host synchronisation
            prefetch(args.model_type, model, upcoming, args.pad_id, args.use_cond2dec, skip_ignored=True)
        model_cost_time += time()
        resolve()                # the previous step's scalars: already on the host
        ys_cond = batch['dconds'].unsqueeze(2).contiguous().view(-1, nprop, 1) if nprop > 0 else None
        ys_mol = batch['trg'][:, 1:].contiguous().view(-1)
        update_cost_time -= time()
        if train:
            optimizer.zero_grad(set_to_none=True)
        loss, RCE_mol, RCE_prop, KLD = loss_function(beta, preds_prop, preds_mol, ys_cond, ys_mol,
                                                     mu, log_var, args.use_cond2dec, args.pad_id)
        scal = torch.stack([RCE_mol.detach(), KLD.detach(), loss.detach()])
        host_scal = ev = None
        if scal.is_cuda:
            # the three scalars leave the device asynchronously, queued BEFORE the backward pass, and are read one step
            # later (resolve(), after the next forward has been queued): the host never stops to wait for a loss; the
            # history and the log lines are the same, one step late
            host_scal = torch.empty(3, dtype=scal.dtype, pin_memory=True)
            host_scal.copy_(scal, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        if train:
            loss.backward()
            optimizer.step()
        update_cost_time += time()
        lr = None
        if args.lr_scheduler == "WarmUpDefault":
            lr = warmup_lr(current_step, args.d_model, args.lr_WarmUpSteps)
        if train and lr is not None:
            for g in optimizer.param_groups:
                g['lr'] = lr
        current_lr = optimizer.param_groups[-1]['lr']
        n_samples += n_onebatch
        if host_scal is not None:
            pending.append((i, n_onebatch, beta, current_lr, host_scal, ev))
        else:
            pending.append((i, n_onebatch, beta, current_lr, scal, None))
            resolve()
    resolve()
    if train and next(model.parameters()).is_cuda:
        from .. import ops
        ops.assert_no_skipped_row_gradients()     # (every earlier step's counter was read with the next step's row plan)
    if train:
        return history, current_step
    return history


def merge_history(history, world_size, device=None):
    """Mean over ranks of the per-step RCE / KLD / LOSS series (what the reference computes by writing one CSV
    per rank and re-reading them on rank 0, trainer1.py:134-151, 237-252): one all-reduce of a [3, steps]
    tensor; BETA and LR are rank-invariant.  Every rank gets the merged history."""
    if world_size <= 1:
        return history
    t = torch.tensor([history['RCE'], history['KLD'], history['LOSS']], dtype=torch.float64)
    if device is not None and dist.get_backend() == 'nccl':
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    t = (t / world_size).cpu().tolist()
    return {'RCE': t[0], 'KLD': t[1], 'LOSS': t[2], 'BETA': list(history['BETA']), 'LR': list(history['LR'])}


def train_model(args, model, optimizer, train_loader, valid_loader, rank, world_size, LOG):
    beta = 0
    current_step = (args.start_epoch - 1) * len(train_loader)
    for epoch in range(args.start_epoch, args.num_epoch + 1):
        if world_size > 1 and hasattr(train_loader, 'set_epoch'):
            train_loader.set_epoch(epoch)
        LOG.info(f'run epoch: {epoch}')
        if args.use_KLA:
            if epoch + 1 >= args.KLA_beg_epoch and beta < args.KLA_max_beta:
                beta = KLAnnealer(epoch, args.KLA_ini_beta, args.KLA_inc_beta, args.KLA_beg_epoch)
        else:
            beta = 1
        LOG.info(f'training start, epoch: {epoch}')
        if world_size > 1:
            dist.barrier()
        model.train()
        train_history, current_step = run_epoch(args, model, optimizer, train_loader, current_step,
                                                beta, LOG, train=True)
        LOG.info('Save training results...')
        sfx = f'_r{rank}' if world_size > 1 else ''
        pd.DataFrame(train_history).to_csv(os.path.join(args.model_folder, f'train_{epoch}{sfx}.csv'))
        LOG.info(f'validation start, epoch: {epoch}')
        if world_size > 1:
            dist.barrier()
        model.eval()
        with torch.no_grad():
            valid_history = run_epoch(args, model, optimizer, valid_loader, current_step, beta, LOG,
                                      train=False)
        LOG.info('Save validation results...')
        pd.DataFrame(valid_history).to_csv(os.path.join(args.model_folder, f'valid_{epoch}{sfx}.csv'))
        if world_size > 1:
            dist.barrier()
        if rank == 0:
            LOG.info('Save model...')
            save_checkpoint(args, model, optimizer, os.path.join(args.model_folder, f'model_{epoch}.pt'))
        if world_size > 1:
            dev = next(model.parameters()).device
            for data_type, his in (('train', train_history), ('valid', valid_history)):
                merged = merge_history(his, world_size, dev)     # same numbers as the reference's CSV round trip
                if rank == 0:
                    pd.DataFrame(merged).to_csv(os.path.join(args.model_folder, f'{data_type}_{epoch}.csv'))
        if world_size > 1:
            dist.barrier()
