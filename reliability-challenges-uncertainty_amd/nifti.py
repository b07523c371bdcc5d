"""Minimal NIfTI-1 single-file codec (.nii / .nii.gz) -- the wire format between the test scripts and
the evaluation scripts of the reference.

The reference writes ``{subject}_probabilities.nii.gz`` (float32), ``{subject}_prediction.nii.gz``
(uint8) and ``{subject}_sigma.nii.gz`` (float32) through SimpleITK
(bin-dl/brats_test_default.py:96-108, bin-dl/brats_test_aleatoric.py:100-110,
bin-dl/isic_test_default.py:106-113) and reads them back in rechun/eval/analysis.py:75-125.
The only properties the path relies on are a lossless float32 / uint8 round trip and the copy of
the image geometry (origin, spacing, direction).  SimpleITK is not available in this image, so this
module implements exactly that subset: header (348 bytes, ``n+1`` magic), sform/qform geometry in
ITK's convention (ITK stores LPS, NIfTI stores RAS: x and y flip sign), gzip transparently.

Array convention as ``sitk.GetArrayFromImage``: numpy index order is ``[z, y, x]`` (x fastest), i.e.
``array.shape == size[::-1]``.
"""
import gzip
import os
import struct
import threading
import zlib

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
           768: np.uint32}
_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}


class ImageProperties:
    """Geometry of an image: the subset of ``pymia.data.conversion.ImageProperties`` the writers copy
    (size, origin, spacing, direction as a flat row-major matrix)."""

    def __init__(self, size, origin=None, spacing=None, direction=None):
        self.size = tuple(int(s) for s in size)                      # (x, y, z)
        n = len(self.size)
        self.origin = tuple(float(o) for o in (origin if origin is not None else (0.0,) * n))
        self.spacing = tuple(float(s) for s in (spacing if spacing is not None else (1.0,) * n))
        self.direction = tuple(float(d) for d in (direction if direction is not None else np.eye(n).reshape(-1)))

    @classmethod
    def from_array(cls, array):
        return cls(array.shape[::-1])

    def __eq__(self, other):
        return (isinstance(other, ImageProperties) and self.size == other.size and
                np.allclose(self.origin, other.origin) and np.allclose(self.spacing, other.spacing) and
                np.allclose(self.direction, other.direction))

    def __repr__(self):
        return 'ImageProperties(size={}, origin={}, spacing={}, direction={})'.format(self.size, self.origin,
                                                                                     self.spacing, self.direction)


def _affine_ras(props):
    """4x4 voxel->world matrix in NIfTI's RAS frame from ITK-style LPS geometry."""
    n = len(props.size)
    d = np.eye(3)
    d[:n, :n] = np.asarray(props.direction, dtype=np.float64).reshape(n, n)
    sp = np.ones(3)
    sp[:n] = props.spacing
    org = np.zeros(3)
    org[:n] = props.origin
    aff = np.eye(4)
    aff[:3, :3] = d * sp[None, :]
    aff[:3, 3] = org
    flip = np.diag([-1.0, -1.0, 1.0, 1.0])     # LPS -> RAS
    return flip @ aff


def _quaternion(rot):
    """(b, c, d, qfac) of a proper/improper 3x3 rotation as NIfTI's qform wants it."""
    r = np.array(rot, dtype=np.float64)
    qfac = 1.0
    if np.linalg.det(r) < 0:
        r[:, 2] = -r[:, 2]
        qfac = -1.0
    tr = 1.0 + r[0, 0] + r[1, 1] + r[2, 2]
    if tr > 0.5:
        a = 0.5 * np.sqrt(tr)
        b = 0.25 * (r[2, 1] - r[1, 2]) / a
        c = 0.25 * (r[0, 2] - r[2, 0]) / a
        d = 0.25 * (r[1, 0] - r[0, 1]) / a
    else:
        xd, yd, zd = 1.0 + r[0, 0] - (r[1, 1] + r[2, 2]), 1.0 + r[1, 1] - (r[0, 0] + r[2, 2]), \
            1.0 + r[2, 2] - (r[0, 0] + r[1, 1])
        if xd > 1.0:
            b = 0.5 * np.sqrt(xd)
            c = 0.25 * (r[0, 1] + r[1, 0]) / b
            d = 0.25 * (r[0, 2] + r[2, 0]) / b
            a = 0.25 * (r[2, 1] - r[1, 2]) / b
        elif yd > 1.0:
            c = 0.5 * np.sqrt(yd)
            b = 0.25 * (r[0, 1] + r[1, 0]) / c
            d = 0.25 * (r[1, 2] + r[2, 1]) / c
            a = 0.25 * (r[0, 2] - r[2, 0]) / c
        else:
            d = 0.5 * np.sqrt(zd)
            b = 0.25 * (r[0, 2] + r[2, 0]) / d
            c = 0.25 * (r[1, 2] + r[2, 1]) / d
            a = 0.25 * (r[1, 0] - r[0, 1]) / d
        if a < 0:
            b, c, d = -b, -c, -d
    return b, c, d, qfac


def write(path, array, properties=None, compresslevel=None):
    """Write ``array`` (numpy ``[z, y, x]`` / ``[y, x]`` order) as NIfTI-1; gzip when the name ends in .gz.
    ``compresslevel``: zlib level of the .gz stream; default 1 for floating-point volumes (probability maps do not deflate:
    ratio 0.91 at every level, level 1 is the fastest way to find that out) and 3 for integer ones (label maps shrink 5-fold at
    level 3 already; Python's default 9 takes seconds on a noisy one).  Readers see the same voxels either way."""
    array = np.ascontiguousarray(array)
    if array.dtype == np.bool_:
        array = array.astype(np.uint8)
    if array.dtype not in _CODES:
        raise ValueError('unsupported dtype {} for NIfTI-1'.format(array.dtype))
    if array.ndim < 2 or array.ndim > 3:
        raise ValueError('only 2-D and 3-D images are supported')
    props = properties if properties is not None else ImageProperties.from_array(array)
    if tuple(props.size) != tuple(array.shape[::-1]):
        raise ValueError('array shape {} does not match image size {}'.format(array.shape, props.size))
    dims = list(array.shape[::-1])
    dim = [array.ndim] + dims + [1] * (7 - len(dims))
    aff = _affine_ras(props)
    sp = np.ones(3)
    sp[:len(props.spacing)] = props.spacing
    b, c, d, qfac = _quaternion(aff[:3, :3] / sp[None, :])
    pixdim = [qfac, sp[0], sp[1], sp[2], 0.0, 0.0, 0.0, 0.0]
    hdr = bytearray(348)
    struct.pack_into('<i', hdr, 0, 348)
    struct.pack_into('<8h', hdr, 40, *dim)
    struct.pack_into('<h', hdr, 70, _CODES[array.dtype])
    struct.pack_into('<h', hdr, 72, array.dtype.itemsize * 8)
    struct.pack_into('<8f', hdr, 76, *pixdim)
    struct.pack_into('<f', hdr, 108, 352.0)           # vox_offset
    struct.pack_into('<f', hdr, 112, 1.0)             # scl_slope
    struct.pack_into('<B', hdr, 123, 2)               # xyzt_units: mm
    struct.pack_into('<h', hdr, 252, 1)               # qform_code: scanner
    struct.pack_into('<h', hdr, 254, 1)               # sform_code
    struct.pack_into('<3f', hdr, 256, b, c, d)
    struct.pack_into('<3f', hdr, 268, *aff[:3, 3])
    struct.pack_into('<4f', hdr, 280, *aff[0])
    struct.pack_into('<4f', hdr, 296, *aff[1])
    struct.pack_into('<4f', hdr, 312, *aff[2])
    hdr[344:348] = b'n+1\x00'
    payload = bytes(hdr) + b'\x00\x00\x00\x00' + array.tobytes()
    if str(path).endswith('.gz'):
        level = compresslevel if compresslevel is not None else (1 if array.dtype.kind == 'f' else 3)
        # mtime 0: the file is a function of the voxels and the geometry alone (a run writes the same bytes whenever, and in
        # whatever order, its writer threads get to it)
        with gzip.GzipFile(path, 'wb', compresslevel=level, mtime=0) as f:
            f.write(payload)
    else:
        with open(path, 'wb') as f:
            f.write(payload)


def _file_bytes(path):
    """The file's bytes, inflated when the name ends in .gz -- every gzip member in ONE zlib call: the GIL is released for the whole stream, so
    the read-ahead threads of the evaluation loop (evalrun._ReadAhead) really inflate side by side.  (``gzip.open(...).read()`` walks the stream in
    8 KB pieces through Python: a few thousand GIL hand-overs per map, which is what four to eight reader threads then spend their time on.)"""
    with open(path, 'rb') as f:
        data = f.read()
    if not str(path).endswith('.gz'):
        return data
    members = []
    while data:
        inflater = zlib.decompressobj(wbits=31)          # 31: a gzip header and trailer are expected (and the CRC checked)
        members.append(inflater.decompress(data))
        if not inflater.eof:
            raise EOFError('{}: compressed file ended before the end-of-stream marker was reached'.format(path))
        data = inflater.unused_data.lstrip(b'\0')       # the next member, if any (zero padding behind a member is legal)
    return members[0] if len(members) == 1 else b''.join(members)


def read(path, dtype=None):
    """-> (numpy array in ``[z, y, x]`` order, ImageProperties).  ``dtype`` casts like ``sitk.ReadImage(path, pixelType)``."""
    raw = _file_bytes(path)
    endian = '<'
    if struct.unpack_from('<i', raw, 0)[0] != 348:
        endian = '>'
        if struct.unpack_from('>i', raw, 0)[0] != 348:
            raise ValueError('{} is not a NIfTI-1 file'.format(path))
    if raw[344:347] != b'n+1':
        raise ValueError('{}: only single-file NIfTI-1 (magic n+1) is supported'.format(path))
    dim = struct.unpack_from(endian + '8h', raw, 40)
    ndim = dim[0]
    code = struct.unpack_from(endian + 'h', raw, 70)[0]
    if code not in _DTYPES:
        raise ValueError('{}: unsupported datatype code {}'.format(path, code))
    pixdim = struct.unpack_from(endian + '8f', raw, 76)
    vox_offset = int(struct.unpack_from(endian + 'f', raw, 108)[0])
    slope, inter = struct.unpack_from(endian + '2f', raw, 112)
    sizes = [d for d in dim[1:1 + ndim]]
    while len(sizes) > 2 and sizes[-1] == 1:
        sizes.pop()
    count = int(np.prod(sizes))
    arr = np.frombuffer(raw, dtype=np.dtype(_DTYPES[code]).newbyteorder(endian), count=count, offset=vox_offset)
    arr = arr.reshape(sizes[::-1]).astype(_DTYPES[code])
    if slope != 0.0 and (slope != 1.0 or inter != 0.0):       # NIfTI-1: scl_slope == 0 means "no scaling", whatever scl_inter holds
        arr = arr * slope + inter
    qform_code, sform_code = struct.unpack_from(endian + '2h', raw, 252)
    n = len(sizes)
    if sform_code > 0:
        aff = np.eye(4)
        aff[0] = struct.unpack_from(endian + '4f', raw, 280)
        aff[1] = struct.unpack_from(endian + '4f', raw, 296)
        aff[2] = struct.unpack_from(endian + '4f', raw, 312)
    elif qform_code > 0:
        # method 2 of the NIfTI-1 standard: rotation from the quaternion (b, c, d), a = sqrt(1 - b^2 - c^2 - d^2), columns scaled by
        # pixdim[1..3] (the third also by qfac = pixdim[0], -1 or 1), translation = the q offsets
        b, c, d = (float(v) for v in struct.unpack_from(endian + '3f', raw, 256))
        a2 = 1.0 - (b * b + c * c + d * d)
        if a2 < 1e-7:                      # |(b, c, d)| >= 1 up to rounding: a 180 degree rotation, renormalise
            norm = 1.0 / np.sqrt(b * b + c * c + d * d)
            a, b, c, d = 0.0, b * norm, c * norm, d * norm
        else:
            a = float(np.sqrt(a2))
        rot = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                        [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                        [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
        qfac = -1.0 if pixdim[0] < 0 else 1.0
        scale = [pixdim[1] or 1.0, pixdim[2] or 1.0, (pixdim[3] or 1.0) * qfac]
        aff = np.eye(4)
        aff[:3, :3] = rot * np.asarray(scale)[None, :]
        aff[:3, 3] = struct.unpack_from(endian + '3f', raw, 268)
    else:
        aff = np.diag([pixdim[1], pixdim[2], pixdim[3], 1.0])
    lps = np.diag([-1.0, -1.0, 1.0, 1.0]) @ aff
    spacing = np.linalg.norm(lps[:3, :3], axis=0)
    spacing[spacing == 0] = 1.0
    direction = lps[:3, :3] / spacing[None, :]
    props = ImageProperties(sizes, lps[:n, 3], spacing[:n], direction[:n, :n].reshape(-1))
    if dtype is not None:
        arr = arr.astype(dtype)
    return arr, props


# ---------------------------------------------------------------------------------- writer hook
# The reference starts one fire-and-forget thread per subject (common/utils/threadhelper.py:7-13).  Here: a pool of writer threads
# (zlib releases the GIL, so they really run beside the test loop) behind a bounded queue -- a test loop that produces subjects
# faster than the disk takes them blocks in do_work instead of piling volumes up in memory.
# (16, not 8: the deterministic / aleatoric scripts -- milliseconds of GPU work and up to three 16 MB maps to deflate per subject -- wait for the
# writers: 0.061 -> 0.037 s per subject, profiles/r05_script_throughput_aleatoric.txt)
WRITER_THREADS = max(2, min(16, (os.cpu_count() or 2) // 2))
_pool = None
_pending = []
_slots = threading.BoundedSemaphore(2 * WRITER_THREADS)


def do_work(fn, *args, in_background=True):
    global _pool
    if not in_background:
        fn(*args)
        return
    if _pool is None:
        import concurrent.futures
        _pool = concurrent.futures.ThreadPoolExecutor(max_workers=WRITER_THREADS, thread_name_prefix='rcu-writer')
    _slots.acquire()

    def run():
        try:
            fn(*args)
        finally:
            _slots.release()

    _pending.append(_pool.submit(run))


def join_all():
    """Wait for every queued write; re-raises the first failure (a fire-and-forget thread would lose it)."""
    done, _pending[:] = list(_pending), []
    for fut in done:
        fut.result()


def argmax_last(probabilities, dtype=np.int64):
    """``np.argmax(probabilities, axis=-1)`` (first maximum wins, a NaN counts as the maximum) as ``dtype``: for two classes one
    compare instead of numpy's generic reduction over a two-element axis."""
    probabilities = np.asarray(probabilities)
    if probabilities.shape[-1] != 2 or probabilities.dtype.kind != 'f':
        return np.argmax(probabilities, axis=-1).astype(dtype, copy=False)
    p0, p1 = probabilities[..., 0], probabilities[..., 1]
    return ((p1 > p0) | ((p1 != p1) & (p0 == p0))).view(np.uint8).astype(dtype, copy=False)


def write_subject(test_dir, subject, probabilities, properties=None, sigma=None, in_background=True, prediction=None):
    """What the reference's ``WriteHook._on_test_subject_end`` does for one assembled subject:
    ``probabilities`` is channel-last ``[..., C]`` float32; writes ``{subject}_probabilities.nii.gz``
    (foreground class), ``{subject}_prediction.nii.gz`` (argmax, uint8) and, for aleatoric runs,
    ``{subject}_sigma.nii.gz`` (sigma of the predicted class; bin-dl/brats_test_aleatoric.py:95-110).
    ``prediction``: the arg-max of ``probabilities`` where the caller has it already."""

    # one job per file: the files of a subject compress side by side, and the argmax stays off the test loop's thread
    def predict():
        if prediction is not None:
            return np.asarray(prediction).astype(np.uint8)
        return argmax_last(probabilities, np.uint8)

    def write_probabilities():
        write(os.path.join(test_dir, '{}_probabilities.nii.gz'.format(subject)),
              np.ascontiguousarray(probabilities[..., 1], dtype=np.float32), properties)

    def write_prediction():
        write(os.path.join(test_dir, '{}_prediction.nii.gz'.format(subject)), predict(), properties)

    def write_sigma():
        sel = np.take_along_axis(sigma, predict()[..., None].astype(np.int64), axis=-1)[..., 0]
        write(os.path.join(test_dir, '{}_sigma.nii.gz'.format(subject)), np.ascontiguousarray(sel, np.float32), properties)

    for job in (write_probabilities, write_prediction) + ((write_sigma,) if sigma is not None else ()):
        do_work(job, in_background=in_background)
