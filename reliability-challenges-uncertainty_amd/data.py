"""Data side of the test scripts: datasets, transforms, collation, loader.

The reference reads BraTS from a pymia HDF5 file and ISIC from a folder of jpg / png files
(common/trainloop/data.py:27-154, common/trainloop/factory.py:10-71, rechun/dl/customdatasets.py:12-112).
pymia and h5py are not available here and data ingestion is outside the hot path, so:

  * ``VolumeDataset`` -- slice-wise access to a directory of ``<subject>.npz`` volumes (``images [D,H,W,C]``,
    ``labels [D,H,W]``, optional geometry).  It yields per slice the entries the reference's extractors
    produce for ``indexing: slice`` (images, subject / slice index, volume shape) and serves the
    ``direct_extractor`` entries (labels, properties, subject name) per subject.
  * ``IsicDataset`` -- the folder dataset of the reference (``<prefix>_Data/*.jpg``,
    ``<prefix>_Part1_GroundTruth/*_segmentation.png``), one sample per subject.
  * the transform registry entries the shipped YAML files use: permute, squeeze, unsqueeze, rescale.
A ``.h5`` dataset path raises with a pointer to ``convert`` instead of failing somewhere inside.
"""
import collections
import glob
import json
import os
import threading
import warnings

import numpy as np
import torch
import torch.utils.data as torch_data

from . import nifti

# The block loader copies out of read-only file mappings through torch.from_numpy (it only reads them); torch warns about the non-writable
# array once per process.  One targeted process-wide filter, installed at import: warnings.catch_warnings() around the copy is not
# thread-safe (it swaps the global filter list) and that copy runs on the loader thread beside the main thread's own warnings.
warnings.filterwarnings('ignore', message='The given NumPy array is not writable', category=UserWarning)


# ------------------------------------------------------------------------------------ transforms
# Every transform also has a ``batched`` form that takes a dict of BLOCKS (``[n, ...]`` arrays: n consecutive samples) and gives, per
# sample, exactly what ``__call__`` gives for that sample alone: VolumeDataset builds a loader batch from a few blocks of a volume
# instead of from 32 per-slice dicts (160 Python-level samples per BraTS subject cost more than the GPU work of the subject).
class Permute:
    def __init__(self, permutation, entries=('images', 'labels')):
        self.permutation, self.entries = tuple(permutation), tuple(entries)

    def __call__(self, sample):
        for e in self.entries:
            if e in sample:
                sample[e] = np.transpose(sample[e], self.permutation)
        return sample

    def batched(self, block):
        for e in self.entries:
            if e in block:
                block[e] = np.transpose(block[e], (0,) + tuple(p % (block[e].ndim - 1) + 1 for p in self.permutation))
        return block


class Squeeze:
    def __init__(self, entries=('images', 'labels'), squeeze_axis=None):
        self.entries, self.axis = tuple(entries), squeeze_axis

    def __call__(self, sample):
        for e in self.entries:
            if e in sample:
                sample[e] = np.squeeze(sample[e], self.axis)
        return sample

    def batched(self, block):
        for e in self.entries:
            if e in block:
                a = block[e]
                if self.axis is None:      # every axis of length one, but never the block axis
                    block[e] = a.reshape((a.shape[0],) + tuple(d for d in a.shape[1:] if d != 1))
                else:
                    axes = self.axis if isinstance(self.axis, (tuple, list)) else (self.axis,)
                    block[e] = np.squeeze(a, tuple(ax % (a.ndim - 1) + 1 for ax in axes))
        return block


class UnSqueeze:
    def __init__(self, axis=-1, entries=('images', 'labels')):
        self.entries, self.axis = tuple(entries), axis

    def __call__(self, sample):
        for e in self.entries:
            if e in sample:
                sample[e] = np.expand_dims(sample[e], self.axis)
        return sample

    def batched(self, block):
        for e in self.entries:
            if e in block:
                a = block[e]
                block[e] = np.expand_dims(a, self.axis % a.ndim + 1)      # position among the sample's ndim + 1 output axes, shifted by the block axis
        return block


class IntensityRescale:
    """min-max rescale of each entry to [lower, upper] (pymia IntensityRescale; parity unpinned)."""

    def __init__(self, lower, upper, entries=('images',)):
        self.lower, self.upper, self.entries = lower, upper, tuple(entries)

    def __call__(self, sample):
        for e in self.entries:
            if e in sample:
                a = sample[e].astype(np.float32)
                lo, hi = a.min(), a.max()
                a = (a - lo) / (hi - lo) if hi > lo else np.zeros_like(a)
                sample[e] = a * (self.upper - self.lower) + self.lower
        return sample

    def batched(self, block):
        for e in self.entries:
            if e in block:
                a = block[e].astype(np.float32)
                flat = a.reshape(a.shape[0], -1)
                shape = (a.shape[0],) + (1,) * (a.ndim - 1)
                lo, hi = flat.min(axis=1).reshape(shape), flat.max(axis=1).reshape(shape)
                span = np.where(hi > lo, hi - lo, np.float32(1))
                a = np.where(hi > lo, (a - lo) / span, np.float32(0)).astype(np.float32)
                block[e] = a * (self.upper - self.lower) + self.lower
        return block


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, sample):
        for t in self.transforms:
            sample = t(sample)
        return sample

    @property
    def batchable(self):
        return all(hasattr(t, 'batched') and (not isinstance(t, Compose) or t.batchable) for t in self.transforms)

    def batched(self, block):
        for t in self.transforms:
            block = t.batched(block)
        return block


transform_registry = {'permute': Permute, 'squeeze': Squeeze, 'unsqueeze': UnSqueeze, 'rescale': IntensityRescale}


def get_transform(params):
    """common/trainloop/factory.py:18-27 for a Parameter or a list of them (None -> identity)."""
    if params is None:
        return Compose([])
    if isinstance(params, (list, tuple)):
        return Compose([get_transform(p) for p in params])
    if params.type not in transform_registry:
        raise ValueError('transform type "{}" unknown'.format(params.type))
    return transform_registry[params.type](**params.params)


# -------------------------------------------------------------------------------------- datasets
def write_volume(dataset_dir, subject, images, labels=None, properties=None):
    """Store one subject of a VolumeDataset: ``images`` float32 ``[D,H,W,C]``, ``labels`` uint8 ``[D,H,W]``."""
    os.makedirs(dataset_dir, exist_ok=True)
    arrays = {'images': np.asarray(images, dtype=np.float32)}
    if labels is not None:
        arrays['labels'] = np.asarray(labels, dtype=np.uint8)
    if properties is not None:
        arrays.update(origin=np.asarray(properties.origin), spacing=np.asarray(properties.spacing),
                      direction=np.asarray(properties.direction))
    np.savez(os.path.join(dataset_dir, subject + '.npz'), **arrays)


def _npz_member_shape(path, name):
    """Shape of array ``name`` of an .npz file from its .npy header alone (np.load(...)[name].shape reads the whole volume)."""
    import zipfile
    with zipfile.ZipFile(path) as z, z.open(name + '.npy') as f:
        version = np.lib.format.read_magic(f)
        reader = np.lib.format.read_array_header_1_0 if version == (1, 0) else np.lib.format.read_array_header_2_0
        return tuple(reader(f)[0])


def _npz_member_mmap(path, name):
    """Array ``name`` of an UNCOMPRESSED .npz file (np.savez: ZIP_STORED members) as a read-only memory map of the file, or None
    when the member is deflated / not a plain array: no read, no CRC pass, no copy -- a slice of the volume is a view of the page
    cache (np.load reads and checksums the whole 63 MB volume for the first slice of a BraTS subject)."""
    import struct
    import zipfile
    try:
        with zipfile.ZipFile(path) as z:
            info = z.getinfo(name + '.npy')
            if info.compress_type != zipfile.ZIP_STORED:
                return None
            header_offset = info.header_offset
        with open(path, 'rb') as f:
            f.seek(header_offset)
            local = f.read(30)
            if local[:4] != b'PK\x03\x04':
                return None
            n_name, n_extra = struct.unpack('<HH', local[26:30])
            f.seek(header_offset + 30 + n_name + n_extra)
            version = np.lib.format.read_magic(f)
            reader = np.lib.format.read_array_header_1_0 if version == (1, 0) else np.lib.format.read_array_header_2_0
            shape, fortran, dtype = reader(f)
            if dtype.hasobject:
                return None
            offset = f.tell()
            # the mapping skips the CRC pass np.load makes over the member: at least the sizes must agree -- a truncated or overwritten
            # file (member shorter than its header says, or reaching behind the end of the file) falls back to np.load, which fails loudly
            nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
            if offset - (header_offset + 30 + n_name + n_extra) + nbytes != info.file_size or offset + nbytes > os.path.getsize(path):
                return None
        return np.memmap(path, dtype=dtype, mode='r', shape=tuple(shape), offset=offset, order='F' if fortran else 'C')
    except (KeyError, OSError, ValueError, struct.error, zipfile.BadZipFile):
        return None


class VolumeDataset(torch_data.Dataset):
    """One sample = one slice (axis 0) of one subject; subjects in sorted order, optionally a subset."""

    def __init__(self, dataset_dir, transform=None, subject_subset=None, slice_categories=('images',), block_loader=True):
        self.slice_categories = tuple(slice_categories)
        self.block_loader = bool(block_loader)      # False: the per-slice path (one __getitem__ per sample) instead of block extraction
        if str(dataset_dir).endswith(('.h5', '.hdf5')):
            raise ValueError('"{}": pymia HDF5 datasets cannot be read in this environment (no h5py / pymia); export the '
                             'volumes with rcu_amd.data.write_volume(<dir>, subject, images, labels) and point '
                             '`dataset:` at that directory'.format(dataset_dir))
        files = sorted(glob.glob(os.path.join(dataset_dir, '*.npz')))
        self.subjects = [os.path.splitext(os.path.basename(f))[0] for f in files]
        if subject_subset is not None:
            keep = set(subject_subset)
            files = [f for f, s in zip(files, self.subjects) if s in keep]
            self.subjects = [s for s in self.subjects if s in keep]
        if not files:
            raise ValueError('no <subject>.npz volumes found in "{}"'.format(dataset_dir))
        self.files = files
        self.transform = transform if transform is not None else Compose([])
        self.index = []          # (subject index, slice index)
        self.shapes = []
        for si, f in enumerate(files):
            shape = _npz_member_shape(f, 'images')
            self.shapes.append(shape)
            self.index.extend((si, k) for k in range(shape[0]))
        # the last few subjects read: the loader thread is on subject i + 1 when the test loop asks for the labels of subject i
        self._cache = collections.OrderedDict()
        self._small = collections.OrderedDict()
        self._cache_lock = threading.Lock()

    def _volume(self, si):
        with self._cache_lock:
            vol = self._cache.get(si)
            if vol is not None:
                return vol
        with np.load(self.files[si]) as z:
            vol = {}
            for k in z.files:
                big = k == 'images'          # the image volume: mapped, not read (labels and geometry are small: read)
                arr = _npz_member_mmap(self.files[si], k) if big else None
                vol[k] = arr if arr is not None else z[k]
        with self._cache_lock:
            self._cache[si] = vol
            while len(self._cache) > 3:
                self._cache.popitem(last=False)
            # labels and geometry of the last subjects, kept apart from the image volumes: the test loop asks for them (direct_extract)
            # several subjects behind the loader thread
            self._small[si] = {k: v for k, v in vol.items() if k != 'images'}
            while len(self._small) > 32:
                self._small.popitem(last=False)
        return vol

    def _subject_entries(self, si):
        with self._cache_lock:
            small = self._small.get(si)
        return small if small is not None else self._volume(si)

    def __len__(self):
        return len(self.index)

    def __getitem__(self, i):
        si, k = self.index[i]
        vol = self._volume(si)
        images = vol['images'][k]
        if isinstance(images, np.memmap):
            images = np.array(images)        # out of the read-only file mapping: a slice the transforms and the collate may own
        sample = {'images': images, 'subject_index': si, 'slice_index': k,
                  'shape': tuple(self.shapes[si][:3]), 'sample_index': i}
        if 'labels' in self.slice_categories:     # extractor `data: {categories: [images, labels]}` (auxiliary_segm)
            sample['labels'] = vol['labels'][k]
        return self.transform(sample)

    def __getitems__(self, indices):
        """A loader batch at once (torch's DataLoader calls this with the batch's indices when it exists): runs of consecutive
        slices of one subject are taken from the volume as blocks and go through the batched forms of the transforms -> a
        ``PreCollated`` batch, equal entry for entry to what CollateDict makes of the per-slice samples."""
        if not getattr(self.transform, 'batchable', False) or not self.block_loader:
            return [self[i] for i in indices]
        keys = ('images', 'labels') if 'labels' in self.slice_categories else ('images',)
        blocks = {k: [] for k in keys}
        meta = {'subject_index': [], 'slice_index': [], 'shape': [], 'sample_index': []}
        b, n = 0, len(indices)
        while b < n:
            si, k0 = self.index[indices[b]]
            e = b + 1
            while e < n and indices[e] == indices[e - 1] + 1 and self.index[indices[e]][0] == si:
                e += 1
            vol = self._volume(si)
            block = self.transform.batched({k: np.asarray(vol[k][k0:k0 + (e - b)]) for k in keys})
            for k in keys:
                blocks[k].append(block[k])
            for i in indices[b:e]:
                meta['subject_index'].append(si)
                meta['slice_index'].append(self.index[i][1])
                meta['shape'].append(tuple(self.shapes[si][:3]))
                meta['sample_index'].append(i)
            b = e
        out = PreCollated()

        def collated(k):
            shape = (n,) + tuple(blocks[k][0].shape[1:])
            if len(shape) == 4 and all(blk.ndim == 4 and blk.transpose(0, 2, 3, 1).flags['C_CONTIGUOUS'] and not blk.flags['C_CONTIGUOUS']
                                       for blk in blocks[k]):
                # The blocks are channel-first VIEWS of the file's channel-last slices (Permute).  The batch keeps the file's memory order -- a
                # plain memcpy per block, straight out of the file mapping -- and is handed over as the channel-first view of it (torch's
                # channels_last memory format: same shape, same values as the stacked per-slice samples); the re-ordering happens on the GPU,
                # behind the upload (steps._images_to_device).  The transposing copy on this thread -- numpy: 12 ms per 32 BraTS slices; torch's
                # OpenMP copy: 3 ms alone, 25 ms beside a busy test loop -- was what the loop waited for with the shipped batch_size 32
                # (tools/loop_timeline.py, round 5).
                base = np.empty((n,) + tuple(blocks[k][0].shape[2:]) + (shape[1],), dtype=blocks[k][0].dtype)
                at = 0
                for blk in blocks[k]:
                    np.copyto(base[at:at + blk.shape[0]], blk.transpose(0, 2, 3, 1))
                    at += blk.shape[0]
                return torch.from_numpy(base).permute(0, 3, 1, 2)
            dst = torch.from_numpy(np.empty(shape, dtype=blocks[k][0].dtype))
            at = 0
            for blk in blocks[k]:       # one (transposing) copy per block, straight out of the file mapping
                piece = dst[at:at + blk.shape[0]]
                if blk.nbytes >= (1 << 20) and all(s_ >= 0 for s_ in blk.strides):
                    # a large channel-last -> channel-first copy: torch splits it over its intra-op threads (numpy's runs on this one
                    # thread: 12 ms for a batch of 32 BraTS slices, beside the loader's other work for it)
                    piece.copy_(torch.from_numpy(blk))      # (a view of the read-only file mapping, only read here: see the filter at the top)
                else:
                    piece.numpy()[...] = blk
                at += blk.shape[0]
            return dst

        out['images'] = collated('images')            # (the key order of the per-slice samples: images, the index entries, labels)
        out.update(meta)
        if 'labels' in keys:
            out['labels'] = collated('labels')
        return out

    def direct_extract(self, subject_index, entries=('labels', 'properties', 'subject')):
        """Per-subject entries of the ``direct_extractor`` list (names, data(labels), files, properties, subject)."""
        vol = self._subject_entries(subject_index)
        out = {}
        if 'labels' in entries and 'labels' in vol:
            out['labels'] = vol['labels']
        if 'properties' in entries:
            d, h, w = self.shapes[subject_index][:3]
            out['properties'] = nifti.ImageProperties((w, h, d), vol.get('origin'), vol.get('spacing'),
                                                      vol.get('direction'))
        if 'subject' in entries:
            out['subject'] = self.subjects[subject_index]
        return out


class IsicDataset(torch_data.Dataset):
    """rechun/dl/customdatasets.py:12-95: ids are the first 12 characters of the file names."""
    LABEL_DIR_POST_FIX = '_Part1_GroundTruth'
    IMAGE_DIR_POST_FIX = '_Data'

    def __init__(self, data_dir_with_task_prefix, transform=None, subject_subset=None, prediction_dir=None):
        from PIL import Image  # noqa: F401  (fail early if PIL is missing)
        self.prediction_dir = prediction_dir
        self.prefix = data_dir_with_task_prefix
        self.transform = transform if transform is not None else Compose([])
        img_dir, label_dir = self.prefix + self.IMAGE_DIR_POST_FIX, self.prefix + self.LABEL_DIR_POST_FIX
        if not (os.path.isdir(img_dir) and os.path.isdir(label_dir)):
            raise ValueError('expected the directories "{}" and "{}"'.format(img_dir, label_dir))
        by_id = {}
        for path in glob.glob(os.path.join(img_dir, '*')) + glob.glob(os.path.join(label_dir, '*')):
            name = os.path.basename(path)
            if name.endswith('_segmentation.png'):
                by_id.setdefault(name[:12], {})['gt'] = path
            elif name.endswith('.jpg'):
                by_id.setdefault(name[:12], {})['image'] = path
        if subject_subset is not None:
            by_id = {k: v for k, v in by_id.items() if k in set(subject_subset)}
        self.files_by_id = {k: v for k, v in by_id.items() if 'gt' in v and 'image' in v}
        if prediction_dir is not None:   # customdatasets.py:33-35, 104-109: `<id>_prediction.nii.gz` of an earlier run
            for path in glob.glob(os.path.join(prediction_dir, '*_prediction.nii.gz')):
                id_ = os.path.basename(path)[:-len('_prediction.nii.gz')]
                if id_ in self.files_by_id:
                    self.files_by_id[id_]['prediction'] = path
            missing = [k for k, v in self.files_by_id.items() if 'prediction' not in v]
            if missing:
                raise ValueError('no prediction file in "{}" for {} subject(s), e.g. {}'.format(prediction_dir, len(missing),
                                                                                              missing[0]))
        self.ids = sorted(self.files_by_id)

    def __len__(self):
        return len(self.ids)

    def get_files_by_id(self, id_):
        f = self.files_by_id[id_]
        return {'image_paths': f['image'], 'label_paths': f['gt']}

    def __getitem__(self, index):
        from PIL import Image
        id_ = self.ids[index]
        f = self.files_by_id[id_]
        sample = {'ids': id_,
                  'labels': np.array(Image.open(f['gt']).convert('L'))[..., np.newaxis].astype(np.uint8),
                  'images': np.array(Image.open(f['image'])).astype(np.float32),
                  'image_paths': f['image'], 'label_paths': f['gt'], 'subject_index': index, 'sample_index': index}
        if self.prediction_dir is not None:   # customdatasets.py:64-69: labels max is 255
            prediction = nifti.read(f['prediction'])[0] * 255
            sample['labels'] = np.concatenate((sample['labels'], prediction[..., np.newaxis].astype(np.uint8)), axis=-1)
        return self.transform(sample)


# ---------------------------------------------------------------------------- collate and loading
class PreCollated(dict):
    """A batch a dataset has collated itself (VolumeDataset.__getitems__)."""


class CollateDict:
    """Stack the tensor entries, keep everything else as per-sample lists (common/data/collate.py:4-16)."""

    def __init__(self, entries=('labels', 'images')):
        self.entries = entries

    def __call__(self, batch):
        if isinstance(batch, PreCollated):
            # the dataset stacked 'images' (and 'labels'): entries this collate would have kept as per-sample lists are unstacked again
            return {key: (list(value) if (torch.is_tensor(value) and key in ('images', 'labels') and key not in self.entries) else value)
                    for key, value in batch.items()}
        out = {}
        for key in batch[0]:
            if key in self.entries:
                out[key] = torch_data.dataloader.default_collate([b[key] for b in batch])
            else:
                out[key] = [b[key] for b in batch]
        return out


class Data:
    def __init__(self, dataset, loader):
        self.dataset = dataset
        self.loader = loader
        self.nb_batches = len(loader)


def load_split(file, k=None):
    """common/data/split.py:84-93."""
    with open(file, 'r') as f:
        d = json.load(f)
    train, valid, test = d['train'], d['valid'], d['test']
    if k is not None:
        train, valid = train[k], valid[k]
        test = [] if test is None else test[k]
    return train, valid, test


def _slice_categories(extractor):
    """Categories of the per-slice `data` extractor entry (default: images only, pymia DataExtractor)."""
    entries = extractor if isinstance(extractor, (list, tuple)) else ([] if extractor is None else [extractor])
    for e in entries:
        if getattr(e, 'type', None) == 'data' and 'categories' in e.params:
            return tuple(e.params['categories'])
    return ('images',)


class BuildVolumeDataset:
    def __call__(self, data_config, **kwargs):
        return VolumeDataset(data_config.dataset, get_transform(data_config.transform), kwargs.get('entries'),
                             _slice_categories(data_config.extractor))


class BuildIsicDataset:
    def __call__(self, data_config, **kwargs):
        return IsicDataset(data_config.dataset, get_transform(data_config.transform), kwargs.get('entries'),
                           kwargs.get('prediction_dir'))


class BuildData:
    """dataset -> sequential (or shuffled) loader with the dict collate (common/trainloop/data.py:140-154)."""

    def __init__(self, build_dataset, **kwargs):
        self.build_dataset = build_dataset
        self.kwargs = kwargs

    def __call__(self, data_config, **kwargs):
        dataset = self.build_dataset(data_config, **{**self.kwargs, **kwargs})
        loader = torch_data.DataLoader(dataset, batch_size=data_config.batch_size, shuffle=bool(data_config.shuffle),
                                       num_workers=0, collate_fn=CollateDict())
        return Data(dataset, loader)
