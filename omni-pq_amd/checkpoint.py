"""Checkpoint format of the reference's training script -- `train.py:153-207`:

    save_checkpoint(args, epoch, model, optimizer, scheduler, save_cur=False, **kwargs)
    load_checkpoint(args, model, optimizer, scheduler, **kwargs)

A checkpoint is ONE `torch.save`d dict {'config': args, 'save_path', 'model', 'optimizer', 'scheduler', 'epoch'[, 'ema_model']};
'model' is the state_dict of the DistributedDataParallel wrapper, so every name carries a `module.` prefix; 'epoch' is an int,
or the strings 'last' / 'best' in the released files (train.py:157-160).  This module reads and writes exactly that, so a file
written by either side loads on the other; the parameter names and shapes of this repo's PQ_Transformer are the reference's
(tests/golden/reference_state_spec.npz).  Differences, all on the tolerant side:
  * the `module.` prefix is added or dropped to match the model being loaded (a bare model can read a DDP checkpoint and
    the other way round);
  * 'ema_model' may hold a state_dict or -- as train.py:192 writes it -- the module itself;
  * `num_batches_tracked` entries of BatchNorm layers are optional (SyncBatchNorm conversion keeps them, older files may not).
"""
import os

import torch

PREFIX = "module."


def _plain(state):
    """state_dict (or module) -> {name without `module.`: tensor}."""
    if isinstance(state, torch.nn.Module):
        state = state.state_dict()
    return {(k[len(PREFIX):] if k.startswith(PREFIX) else k): v for k, v in state.items()}


def _load_into(module, state):
    plain = _plain(state)
    target = module.state_dict()
    wants_prefix = any(k.startswith(PREFIX) for k in target)
    fixed = {((PREFIX + k) if wants_prefix else k): v for k, v in plain.items()}
    missing = [k for k in target if k not in fixed and not k.endswith("num_batches_tracked")]
    unexpected = [k for k in fixed if k not in target]
    if missing or unexpected:
        raise RuntimeError(f"checkpoint does not match the model: missing {missing[:5]} ({len(missing)}), "
                           f"unexpected {unexpected[:5]} ({len(unexpected)})")
    module.load_state_dict(fixed, strict=False)


def _read(path, trust_pickle):
    """The reference's checkpoints hold an argparse.Namespace (`config`) next to tensors, and some an entire pickled EMA
    module.  Try the safe loader first (tensors, containers, Namespace); only an explicit `trust_pickle=True` falls
    back to full unpickling -- which executes code from the file."""
    import argparse
    import pickle
    try:
        with torch.serialization.safe_globals([argparse.Namespace]):
            return torch.load(path, map_location='cpu', weights_only=True)
    except pickle.UnpicklingError as safe_err:      # what a weights_only rejection raises; I/O errors and corrupt archives propagate
        if not trust_pickle:
            raise RuntimeError(
                f"{path}: not loadable with weights_only=True ({type(safe_err).__name__}: {safe_err}).  If the file "
                "comes from a source you trust (e.g. it pickles a whole EMA module, train.py:181-207), pass "
                "trust_pickle=True to load_checkpoint") from safe_err
        return torch.load(path, map_location='cpu', weights_only=False)


def load_checkpoint(args, model, optimizer, scheduler, trust_pickle=False, **kwargs):
    """Restore model / optimizer / scheduler (and `ema_model=` when args.ema) from args.checkpoint_path; sets
    args.start_epoch to the epoch after the saved one ('last' counts as 600, 'best' as 0, as in the reference).
    A state the file holds but no object was given for (optimizer=None / scheduler=None) is skipped with a warning."""
    import warnings
    checkpoint = _read(args.checkpoint_path, trust_pickle)
    epoch = checkpoint['epoch']
    epoch = {'last': 600, 'best': 0}.get(epoch, epoch)
    args.start_epoch = epoch + 1
    _load_into(model, checkpoint['model'])
    for name, obj in (('optimizer', optimizer), ('scheduler', scheduler)):
        if obj is not None and checkpoint.get(name) is not None:
            obj.load_state_dict(checkpoint[name])
        elif checkpoint.get(name) is not None:
            warnings.warn(f"load_checkpoint: the file holds a {name} state but no {name} was given; it is NOT restored")
        elif obj is not None:
            warnings.warn(f"load_checkpoint: no {name} state in the file; the given {name} keeps its own")
    if getattr(args, 'ema', False) and 'ema_model' in kwargs:
        _load_into(kwargs['ema_model'], checkpoint['ema_model'] if 'ema_model' in checkpoint else checkpoint['model'])
    return epoch


def save_checkpoint(args, epoch, model, optimizer, scheduler, save_cur=False, **kwargs):
    """Write args.log_dir/ckpt_epoch_{epoch}.pth when `save_cur` or epoch % args.save_freq == 0 (else nothing, as in the
    reference).  -> the path written, or None."""
    if not (save_cur or epoch % args.save_freq == 0):
        return None
    path = os.path.join(args.log_dir, f'ckpt_epoch_{epoch}.pth')
    state = {
        'config': args,
        'save_path': path,
        'model': model.state_dict(),
        'optimizer': optimizer.state_dict(),
        'scheduler': scheduler.state_dict(),
        'epoch': epoch,
    }
    if getattr(args, 'ema', False) and 'ema_model' in kwargs:
        ema = kwargs['ema_model']
        state['ema_model'] = ema.state_dict() if isinstance(ema, torch.nn.Module) else ema
    torch.save(state, path)
    return path
