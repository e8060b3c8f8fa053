"""Sample / reference .npz I/O in the reference's formats (SURVEY.md §8b "File formats").

Reference .npz files written here load with plain `np.load(..., allow_pickle=True)` exactly like
the reference's, but the large arrays (indexes / distances / null_ratios, ~0.7 GB at 15 kb) are
stored uncompressed inside the zip: single-thread zlib on them costs ~1 min per write in the
reference (np.savez_compressed, newref_control.py:145,176,237) and would dominate the GPU run.
"""
import io
import zipfile

import numpy as np

_BIG = 8 << 20


def save_npz(path, arrays, compress_small=True):
    if not str(path).endswith(".npz"):
        path = str(path) + ".npz"
    with zipfile.ZipFile(path, "w", allowZip64=True) as zf:
        for name, val in arrays.items():
            arr = np.asanyarray(val)
            buf = io.BytesIO()
            np.lib.format.write_array(buf, arr, allow_pickle=True)
            data = buf.getbuffer()
            ctype = zipfile.ZIP_DEFLATED if (compress_small and arr.nbytes < _BIG) else zipfile.ZIP_STORED
            zf.writestr(zipfile.ZipInfo(name + ".npy"), data, compress_type=ctype)
    return path


def load_sample(path):
    """-> (sample dict "1".."24" -> int32 array, binsize) as written by `convert`
    (main.py:33-35, convert_tools.py:110-119)."""
    npz = np.load(path, encoding="latin1", allow_pickle=True)
    return npz["sample"].item(), int(npz["binsize"])


def save_sample(path, sample, binsize, quality=None):
    np.savez_compressed(path, binsize=binsize, sample=sample, quality=quality or {})


def load_reference(path):
    npz = np.load(path, encoding="latin1", allow_pickle=True)
    return {k: npz[k] for k in npz.files}
