"""Configuration surface: the reference's YAML / JSON test configurations, loaded unchanged.

Mirrors common/trainloop/config.py:44-190 and common/configuration/config.py:4-37 (which build on pymia's
``ConfigurationBase`` -- absent here, parity pinned de facto by the reference's 66 YAML files):

    config:
      test_name / test_dir / model_dir / split / seed / test_at
      others: {free-form, e.g. mc: 20, model_dir: [...]}
      test_data: {batch_size, dataset, num_workers, shuffle, extractor, transform, indexing, direct_extractor, ...}
    meta: {type: test-config, version: 0}

Entries of the "parametric" members (transform, extractor, indexing, ...) use the single-key-dict idiom
``{type: {params}}`` or a bare ``type`` string and become ``Parameter`` objects.
"""
import json
import os

import yaml


class Parameter:
    """``{type: params}`` (DictableParameterExt, common/configuration/config.py:21-37)."""

    def __init__(self, type_=None, **params):
        self.type = type_
        self.params = params

    @classmethod
    def parse(cls, entry):
        if isinstance(entry, str):
            return cls(entry)
        if isinstance(entry, dict):
            if len(entry) != 1:
                raise ValueError('a parametric entry must be a single-key dict, got {}'.format(entry))
            type_, params = next(iter(entry.items()))
            return cls(type_, **(params or {}))
        raise ValueError('cannot parse parametric entry {!r}'.format(entry))

    def to_dict(self):
        return {self.type: self.params} if self.params else self.type

    def __repr__(self):
        return 'Parameter({!r}, {!r})'.format(self.type, self.params)


def _parse_parametric(value):
    if value is None:
        return None
    if isinstance(value, list):
        return [Parameter.parse(v) for v in value]
    return Parameter.parse(value)


def _dump_parametric(value):
    if value is None:
        return None
    if isinstance(value, list):
        return [v.to_dict() for v in value]
    return value.to_dict()


class OtherParameters:
    """Free-form bag: unknown keys are attached verbatim (config.py:110-121)."""
    _parametric = ('model', 'transform', 'additional_models', 'additional_optimizers')

    def from_dict(self, d):
        for k, v in (d or {}).items():
            setattr(self, k, _parse_parametric(v) if k in self._parametric else v)
        return self

    def to_dict(self):
        return {k: (_dump_parametric(v) if k in self._parametric else v) for k, v in vars(self).items()}


class DataConfiguration:
    _parametric = ('extractor', 'transform', 'indexing', 'selection_strategy', 'selection_extractor',
                   'direct_extractor', 'direct_transform')

    def __init__(self):
        self.dataset = ''
        self.batch_size = 10
        self.num_workers = 1
        self.extractor = None
        self.transform = None
        self.indexing = None
        self.selection_strategy = None
        self.selection_extractor = None
        self.shuffle = True
        self.direct_extractor = None
        self.direct_transform = None
        self.others = OtherParameters()

    def from_dict(self, d):
        for k, v in (d or {}).items():
            if k == 'others':
                self.others = OtherParameters().from_dict(v)
            elif k in self._parametric:
                setattr(self, k, _parse_parametric(v))
            else:
                setattr(self, k, v)
        return self

    def to_dict(self):
        out = {}
        for k, v in vars(self).items():
            if k == 'others':
                out[k] = v.to_dict()
            elif k in self._parametric:
                out[k] = _dump_parametric(v)
            else:
                out[k] = v
        return out


class TestConfiguration:
    __test__ = False   # not a pytest class
    TYPE, VERSION = 'test-config', 0

    def __init__(self):
        self.seed = 20
        self.split = ''
        self.model_dir = ''
        self.test_name = ''
        self.test_dir = None
        self.test_at = ''            # 'best', 'last' or an epoch number
        self.test_data = DataConfiguration()
        self.others = OtherParameters()

    def from_dict(self, d):
        for k, v in (d or {}).items():
            if k == 'test_data':
                self.test_data = DataConfiguration().from_dict(v)
            elif k == 'others':
                self.others = OtherParameters().from_dict(v)
            else:
                setattr(self, k, v)
        return self

    def to_dict(self):
        out = dict(vars(self))
        out['test_data'] = self.test_data.to_dict()
        out['others'] = self.others.to_dict()
        return out


def _read(path):
    with open(path, 'r') as f:
        if path.endswith('.json'):
            return json.load(f)
        return yaml.safe_load(f)


def load(path, config_cls=TestConfiguration):
    """File with ``config`` and ``meta`` sections -> configuration object; the meta type must match."""
    d = _read(path)
    meta = d.get('meta', {})
    if meta.get('type') != config_cls.TYPE:
        raise ValueError('configuration "{}" has type "{}" (expected "{}")'.format(path, meta.get('type'), config_cls.TYPE))
    return config_cls().from_dict(d.get('config', {}))


def save(path, config):
    d = {'config': config.to_dict(), 'meta': {'type': config.TYPE, 'version': config.VERSION}}
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, 'w') as f:
        if path.endswith('.json'):
            json.dump(d, f, indent=2)
        else:
            yaml.safe_dump(d, f, default_flow_style=False)
