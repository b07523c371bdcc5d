"""Model directory layout, checkpoint discovery and loading (the input side of the test scripts).

Mirrors common/model/management.py:14-161:
    <train_dir>/model_<id>/model.json                      {"model": {"type": "unet", "params": {...}}, "optimizer": ...}
    <train_dir>/model_<id>/checkpoints/checkpoint_epNNN.pth, checkpoint_epNNN-best.pth
    checkpoint = torch.load(...) -> {'state_dict', 'epoch', 'optimizer', 'best_score'}
``test_at`` is 'best', 'last' or an epoch number.  The model type is resolved through
rcu_amd.model.model_registry, i.e. 'unet' builds the HIP-backed drop-in.
"""
import glob
import json
import os

import torch

from . import model as model_mod


class ModelFiles:
    CHECKPOINT_PLACEHOLDER = 'checkpoint{postfix}_ep{epoch:03d}.pth'
    BEST_PLACEHOLDER = 'checkpoint{postfix}_ep{epoch:03d}-best.pth'
    MODELDIR_PREFIX = 'model_'

    def __init__(self, root_model_dir: str, identifier: str) -> None:
        self.identifier = identifier
        self.root_model_dir = root_model_dir

    @classmethod
    def from_model_dir(cls, model_dir: str):
        model_dir = model_dir.rstrip('/')
        return cls(os.path.dirname(model_dir), os.path.basename(model_dir)[len(cls.MODELDIR_PREFIX):])

    @property
    def model_dir(self) -> str:
        return os.path.join(self.root_model_dir, self.MODELDIR_PREFIX + self.identifier)

    @property
    def weight_checkpoint_dir(self) -> str:
        return os.path.join(self.model_dir, 'checkpoints')

    def model_path(self, postfix='') -> str:
        return os.path.join(self.model_dir, 'model{}.json'.format('-' + postfix if postfix else ''))

    def build_checkpoint_path(self, epoch: int, is_best=False, postfix=''):
        pattern = self.BEST_PLACEHOLDER if is_best else self.CHECKPOINT_PLACEHOLDER
        return os.path.join(self.weight_checkpoint_dir, pattern.format(epoch=epoch, postfix='-' + postfix if postfix else ''))


def find_best_checkpoint_epoch(checkpoint_dir):
    hits = glob.glob(os.path.join(checkpoint_dir, 'checkpoint*ep*-best.pth'))
    if not hits:
        return None
    tail = len('-best.pth')
    return int(os.path.basename(hits[0])[-tail - 3:-tail])


def find_last_checkpoint_epoch(checkpoint_dir):
    hits = glob.glob(os.path.join(checkpoint_dir, 'checkpoint*ep' + 3 * '[0-9]' + '.pth'))
    if not hits:
        return None
    return max(int(os.path.basename(h)[-len('.pth') - 3:-len('.pth')]) for h in hits)


def find_checkpoint_file(checkpoint_dir, epoch_or_best_or_last, postfix=''):
    if not isinstance(epoch_or_best_or_last, (str, int)):
        raise AttributeError('Expected epoch_or_best_or_last types are (string, int), not {}'
                             .format(type(epoch_or_best_or_last)))
    epoch, best = epoch_or_best_or_last, ''
    if isinstance(epoch, str):
        if epoch == 'last':
            epoch = find_last_checkpoint_epoch(checkpoint_dir)
        elif epoch == 'best':
            epoch, best = find_best_checkpoint_epoch(checkpoint_dir), '-best'
        else:
            raise ValueError("allowed string values for epoch are ('last', 'best')")
    if epoch is None:
        return None
    hits = glob.glob(os.path.join(checkpoint_dir, 'checkpoint*ep*{:03d}{}.pth'.format(epoch, best)))
    prefix = 'checkpoint' + ('-' + postfix if postfix else '')
    hits = [h for h in hits if os.path.basename(h).startswith(prefix)]
    return hits[0] if hits else None


def load_model_from_parameters(model_path):
    """model.json -> model instance (management.py:66-87)."""
    if not os.path.exists(model_path):
        raise ValueError('missing model file {}'.format(model_path))
    with open(model_path, 'r') as f:
        d = json.load(f)
    spec = d['model']
    if spec['type'] not in model_mod.model_registry:
        raise ValueError('model type "{}" unknown'.format(spec['type']))
    return model_mod.model_registry[spec['type']](**spec.get('params', {}))


def load_checkpoint(checkpoint_path, model):
    """management.py:56-64; returns the remaining checkpoint entries (epoch, best_score, ...)."""
    if checkpoint_path is None or not os.path.exists(checkpoint_path):
        raise ValueError('missing checkpoint file {}'.format(checkpoint_path))
    checkpoint = torch.load(checkpoint_path, map_location='cpu')
    model.load_state_dict(checkpoint.pop('state_dict'))
    checkpoint.pop('optimizer', None)
    return checkpoint


def save_model(model_files: ModelFiles, model_type, params, state_dict, epoch=1, is_best=True, best_score=None):
    """Write model.json + one checkpoint in the reference's layout (what its training hooks produce,
    common/trainloop/hooks.py:297-328) -- used to package synthetic weights for the drop-in scripts."""
    os.makedirs(model_files.weight_checkpoint_dir, exist_ok=True)
    with open(model_files.model_path(), 'w') as f:
        json.dump({'model': {'type': model_type, 'params': params}, 'optimizer': {'type': 'adam', 'params': {}}}, f)
    path = model_files.build_checkpoint_path(epoch, is_best=is_best)
    torch.save({'state_dict': state_dict, 'epoch': epoch, 'optimizer': {}, 'best_score': best_score}, path)
    return path
