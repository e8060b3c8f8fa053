"""Sample / reference .npz I/O in the reference's formats (SURVEY.md §8b "File formats").

Reference .npz files written here load with plain `np.load(..., allow_pickle=True)` exactly like
the reference's (`newref_control.py:145,176,237` write them with np.savez_compressed), but the
large arrays (indexes / distances / null_ratios, 2.5 GB at 15 kb) are STORED, not deflated:
single-thread zlib on them costs minutes.  Writing and reading them is the biggest item of the
CLI's wall-clock once the search runs on the GPU, so both go around `zipfile` for the big members:

  save_npz        lays the archive out up front (stored members have known sizes), computes the
                  CRC-32 of the big members on worker threads (zlib releases the GIL) and writes
                  them with positional writes from those threads; small members are deflated.
                  The result is an ordinary ZIP (ZIP64 records when needed).
  load_reference  reads stored `.npy` members straight into their arrays (readinto from worker
                  threads); their CRC-32 is checked like zipfile would -- per chunk on the reader
                  threads, combined with crc32_combine (WCX_NPZ_NO_CRC=1 skips it) -- and their
                  sizes against the file; anything else goes through np.load.
"""
import io
import os
import struct
import threading
import tempfile
import zipfile
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_UMASK = os.umask(0)
os.umask(_UMASK)

_BIG = 8 << 20            # stored by the direct writer / read by the direct reader
_DEFLATE_MAX = 1 << 20    # anything larger is mostly incompressible doubles: stored as is


def _cpu_budget():
    """CPUs this process may really use: the cgroup quota when there is one (a container sees every
    core of the host in os.cpu_count()), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


_THREADS = max(2, min(16, _cpu_budget()))
_LOAD_THREADS = max(2, min(32, _cpu_budget()))      # sample import: inflate + unpickle, mostly GIL-free
_Z64 = 0xFFFFFFFF           # sizes / offsets from here on go into ZIP64 extra fields
_DOS_TIME, _DOS_DATE = 0, (1980 - 1980) << 9 | 1 << 5 | 1      # 1980-01-01 00:00, like np.savez


def _npy_header(arr):
    """The bytes np.save puts before the raw data of `arr` (format 1.0, or 2.0 for huge headers)."""
    d = np.lib.format.header_data_from_array_1_0(arr)
    buf = io.BytesIO()
    try:
        np.lib.format.write_array_header_1_0(buf, d)
    except ValueError:
        buf = io.BytesIO()
        np.lib.format.write_array_header_2_0(buf, d)
    return buf.getvalue()


def _raw_view(arr):
    """The array's bytes in the order the header announces (C or Fortran), without a copy."""
    if arr.flags.c_contiguous:
        return memoryview(arr).cast("B")
    return memoryview(arr.T).cast("B")          # F-contiguous: header says fortran_order=True


def _crc_chunks(mv, nparts):
    """Chunk boundaries of a byte view for parallel work."""
    n = len(mv)
    step = max(1 << 20, -(-n // nparts))
    return [(o, min(o + step, n)) for o in range(0, n, step)]


class PrefixConst:
    """A table whose first `n_prefix` rows all hold one value and whose remaining rows are `tail` -- what a
    gonosomal pass produces: get_reference fills the autosomal target rows of indexes / distances with 0 / 1
    (newref_tools.py:186-191), 1.3 GB of the 2.5 GB of a 15 kb reference.  NpzWriter.add stores such a member
    deflated -- the constant rows as a few hundred KB of pre-built deflate blocks, the tail in stored blocks --
    without the rows ever existing on the host; np.load inflates it to the same array as ever, and
    load_reference rebuilds it from the fill value and the tail.  np.asarray() materialises it."""

    def __init__(self, n_prefix, fill, tail):
        self.tail = np.ascontiguousarray(tail)
        self.n_prefix = int(n_prefix)
        self.dtype = self.tail.dtype
        self.fill = self.dtype.type(fill)
        self.shape = (self.n_prefix + self.tail.shape[0],) + self.tail.shape[1:]
        self.ndim = len(self.shape)

    def __len__(self):
        return self.shape[0]

    @property
    def row_bytes(self):
        return int(np.prod(self.shape[1:], dtype=np.int64)) * self.dtype.itemsize

    @property
    def nbytes(self):
        return self.shape[0] * self.row_bytes

    def materialize(self):
        full = np.empty(self.shape, dtype=self.dtype)
        full[:self.n_prefix] = self.fill
        full[self.n_prefix:] = self.tail
        return full

    def __array__(self, dtype=None, copy=None):
        full = self.materialize()
        return full if dtype is None else full.astype(dtype, copy=False)

    def __getitem__(self, item):
        return self.materialize()[item]


_HYB_ID = 0x4357                     # ZIP extra-field id of a hybrid member's description ("WC")
_HYB_FMT = "<4sHQQQQ8s"              # magic, itemsize, header bytes, prefix bytes, tail offset, tail bytes, fill
_HYB_CHUNK = 1 << 20                 # constant bytes per pre-built deflate piece
_SB_MAX = 65535                      # payload of one stored deflate block


def _stored_blocks(view, last_final):
    """`view` as stored deflate blocks (5-byte header + <= 65535 bytes each); the last one carries BFINAL
    if last_final.  An empty view with last_final gives one empty final block."""
    src = np.frombuffer(view, dtype=np.uint8)
    n = len(src)
    n_full, rem = divmod(n, _SB_MAX)
    n_blocks = n_full + (1 if rem or (n == 0 and last_final) else 0)
    out = np.empty(n + 5 * n_blocks, dtype=np.uint8)
    head_full = np.frombuffer(struct.pack("<BHH", 0, _SB_MAX, 0), dtype=np.uint8)
    if n_full:
        blk = out[:n_full * (_SB_MAX + 5)].reshape(n_full, _SB_MAX + 5)
        blk[:, :5] = head_full
        blk[:, 5:] = src[:n_full * _SB_MAX].reshape(n_full, _SB_MAX)
    if n_blocks > n_full:
        o = n_full * (_SB_MAX + 5)
        out[o:o + 5] = np.frombuffer(struct.pack("<BHH", 0, rem, rem ^ 0xFFFF), dtype=np.uint8)
        out[o + 5:] = src[n_full * _SB_MAX:]
    if last_final and n_blocks:
        out[(n_blocks - 1) * (_SB_MAX + 5)] = 1
    return out


def _unstore_blocks(buf, nbytes, dst):
    """The inverse: `buf` = the stored blocks of `nbytes` payload bytes (last one final) -> dst (uint8 view).
    Raises IOError if a block header is not what _stored_blocks writes."""
    n_full, rem = divmod(nbytes, _SB_MAX)
    n_blocks = n_full + (1 if rem or nbytes == 0 else 0)
    if len(buf) != nbytes + 5 * n_blocks:
        raise IOError("stored-block region has {} bytes, expected {}".format(len(buf), nbytes + 5 * n_blocks))
    src = np.frombuffer(buf, dtype=np.uint8)
    heads = []
    if n_full:
        blk = src[:n_full * (_SB_MAX + 5)].reshape(n_full, _SB_MAX + 5)
        dst[:n_full * _SB_MAX].reshape(n_full, _SB_MAX)[:] = blk[:, 5:]
        heads.append(blk[:, :5])
    if n_blocks > n_full:
        o = n_full * (_SB_MAX + 5)
        dst[n_full * _SB_MAX:] = src[o + 5:]
        heads.append(src[o:o + 5].reshape(1, 5))
    h = np.concatenate(heads)
    lens = h[:, 1].astype(np.int64) | (h[:, 2].astype(np.int64) << 8)
    nlens = h[:, 3].astype(np.int64) | (h[:, 4].astype(np.int64) << 8)
    want = np.full(n_blocks, _SB_MAX, dtype=np.int64)
    if n_blocks > n_full:
        want[-1] = rem
    final = np.zeros(n_blocks, dtype=np.uint8)
    final[-1] = 1
    if not (np.array_equal(lens, want) and np.array_equal(nlens, want ^ 0xFFFF) and np.array_equal(h[:, 0], final)):
        raise IOError("damaged stored-block header")


def _crc_repeat(crc, length, times):
    """CRC-32 of `times` copies of a block with CRC `crc` and `length` bytes (binary powers of crc32_combine)."""
    out, cur, cur_len = 0, crc, length
    while times:
        if times & 1:
            out = crc32_combine(out, cur, cur_len)
        times >>= 1
        if times:
            cur = crc32_combine(cur, cur, cur_len)
            cur_len *= 2
    return out


def _const_crc(pattern, nbytes):
    """CRC-32 of `nbytes` bytes of the repeated `pattern` (what _const_region computes beside its blocks)."""
    q, r = divmod(nbytes, _HYB_CHUNK)
    crc = _crc_repeat(zlib.crc32(pattern * (_HYB_CHUNK // len(pattern))), _HYB_CHUNK, q) if q else 0
    return crc32_combine(crc, zlib.crc32(pattern * (r // len(pattern))), r) if r else crc


def _hybrid_info(extra):
    """The description _add_hybrid left in a member's ZIP extra field, or None."""
    o = 0
    while o + 4 <= len(extra):
        hid, n = struct.unpack_from("<HH", extra, o)
        if hid == _HYB_ID and n == struct.calcsize(_HYB_FMT):
            magic, itemsize, head_len, prefix_bytes, tail_off, tail_bytes, fill = struct.unpack_from(_HYB_FMT, extra, o + 4)
            if magic == b"wcx1":
                return {"itemsize": itemsize, "head_len": head_len, "prefix_bytes": prefix_bytes,
                        "tail_off": tail_off, "tail_bytes": tail_bytes, "fill": fill[:itemsize]}
        o += 4 + n
    return None


def _const_region(pattern, nbytes):
    """(raw deflate blocks -- none final, byte aligned --, CRC-32) of `nbytes` bytes of the repeated
    `pattern`: one pre-built piece of _HYB_CHUNK bytes, repeated, + one for the remainder (full flushes:
    every piece is self-contained)."""
    assert _HYB_CHUNK % len(pattern) == 0 and nbytes % len(pattern) == 0
    q, r = divmod(nbytes, _HYB_CHUNK)

    def piece(n):
        data = pattern * (n // len(pattern))
        co = zlib.compressobj(zlib.Z_DEFAULT_COMPRESSION, zlib.DEFLATED, -15)
        return co.compress(data) + co.flush(zlib.Z_FULL_FLUSH), zlib.crc32(data)
    comp, crc = b"", 0
    if q:
        zc, cc = piece(_HYB_CHUNK)
        comp = zc * q
        crc = _crc_repeat(cc, _HYB_CHUNK, q)
    if r:
        zr, cr = piece(r)
        comp += zr
        crc = crc32_combine(crc, cr, r)
    return comp, crc


class NpzWriter:
    """A .npz written member by member: add() plans the member, fixes its place in the file and hands
    its payload (CRC-32 + positional writes, in pieces) to the worker threads at once, so a table can
    be on its way to the file while the next one is still being computed; close() writes the local
    headers and the central directory, syncs and renames.  The file appears under its final name only
    after EVERY write has succeeded (a failed write -- disk full -- must not leave a plausible-looking
    archive behind); the temporary name is unique (two writers of one target, or somebody's stale
    .tmp, must not trample each other).  Arrays handed to add() must stay unchanged until their
    payload is written; the writer lets go of a member's array as soon as that is the case
    (flush_async), so a caller that drops its own reference gets gigabytes of tables freed on the
    writer's threads beside its next work instead of at the end."""

    def __init__(self, path, compress_small=True):
        if not str(path).endswith(".npz"):
            path = str(path) + ".npz"
        self.final_path = str(path)
        self.compress_small = compress_small
        self.members, self.jobs, self.flushes = [], [], []
        self.lock = threading.Lock()
        self.off = 0
        self.fd, self.tmp_path = tempfile.mkstemp(dir=os.path.dirname(os.path.abspath(self.final_path)) or ".",
                                                  prefix=os.path.basename(self.final_path) + ".", suffix=".tmp")
        os.fchmod(self.fd, 0o644 & ~_UMASK)
        self.ex = ThreadPoolExecutor(max_workers=_THREADS)
        self.closed = False

    def _add_hybrid(self, name, pc):
        """A PrefixConst as ONE deflated member: [stored block: .npy header][pre-built blocks: the constant
        rows][stored blocks: the tail, the last one final] + a description in the ZIP extra field (which
        np.load / zipfile skip) so that load_reference can rebuild it without inflating."""
        fname = (name + ".npy").encode("utf-8")
        shell = np.lib.stride_tricks.as_strided(np.empty(1, dtype=pc.dtype), shape=pc.shape,
                                                strides=(0,) * pc.ndim)         # (shape + dtype for the header)
        d = np.lib.format.header_data_from_array_1_0(shell)
        d["fortran_order"] = False
        buf = io.BytesIO()
        try:
            np.lib.format.write_array_header_1_0(buf, d)
        except ValueError:
            buf = io.BytesIO()
            np.lib.format.write_array_header_2_0(buf, d)
        head = buf.getvalue()
        pattern = pc.fill.tobytes()
        prefix_bytes = pc.n_prefix * pc.row_bytes
        zc, crc_p = _const_region(pattern, prefix_bytes)
        tail = memoryview(pc.tail.reshape(-1).view(np.uint8))
        front = _stored_blocks(memoryview(head), False).tobytes() + zc
        comp = np.concatenate([np.frombuffer(front, dtype=np.uint8), _stored_blocks(tail, True)])
        crc = crc32_combine(crc32_combine(zlib.crc32(head), crc_p, prefix_bytes), zlib.crc32(tail), len(tail))
        xtra = struct.pack(_HYB_FMT, b"wcx1", pc.dtype.itemsize, len(head), prefix_bytes, len(front),
                           len(tail), pattern.ljust(8, b"\0"))
        m = {"name": fname, "head": b"", "raw": memoryview(comp), "method": 8,
             "usize": len(head) + prefix_bytes + len(tail), "crc": crc,
             "xtra": struct.pack("<HH", _HYB_ID, len(xtra)) + xtra}
        self._place(m, None)

    def add(self, name, val):
        if isinstance(val, PrefixConst):
            if (val.n_prefix * val.row_bytes >= _BIG and val.dtype.itemsize in (1, 2, 4, 8)
                    and not val.dtype.hasobject):
                return self._add_hybrid(name, val)
            val = val.materialize()
        arr = np.asanyarray(val)
        fname = (name + ".npy").encode("utf-8")
        big = (arr.nbytes >= _BIG and arr.dtype != object and not arr.dtype.hasobject
               and (arr.flags.c_contiguous or arr.flags.f_contiguous))
        if big:
            m = {"name": fname, "head": _npy_header(arr), "raw": _raw_view(arr), "method": 0}
        else:
            buf = io.BytesIO()
            np.lib.format.write_array(buf, arr, allow_pickle=True)
            data = buf.getvalue()
            if self.compress_small and len(data) < _DEFLATE_MAX:
                co = zlib.compressobj(zlib.Z_DEFAULT_COMPRESSION, zlib.DEFLATED, -15)
                comp = co.compress(data) + co.flush()
                m = {"name": fname, "head": b"", "raw": memoryview(comp), "method": 8,
                     "usize": len(data), "crc": zlib.crc32(data)}
            else:
                m = {"name": fname, "head": b"", "raw": memoryview(data), "method": 0}
        self._place(m, arr if big else None)

    def _place(self, m, keep):
        # layout (stored members have known sizes: nothing depends on the CRCs yet)
        m.setdefault("usize", len(m["head"]) + len(m["raw"]))
        m.setdefault("xtra", b"")
        m["csize"] = len(m["head"]) + len(m["raw"])
        m["offset"] = self.off
        m["z64"] = m["usize"] >= _Z64 or m["csize"] >= _Z64
        m["lhdr_len"] = 30 + len(m["name"]) + (20 if m["z64"] else 0) + len(m["xtra"])
        m["data_off"] = self.off + m["lhdr_len"]
        self.off = m["data_off"] + m["csize"]
        self.members.append(m)
        # every chunk of a stored member: CRC-32 and positional write on one worker (zlib and
        # os.pwrite release the GIL); the member's CRC is the chunks' combined in order
        if "crc" in m:
            fs = [self.ex.submit(_pwrite_all, self.fd, m["raw"], m["data_off"])]
        else:
            base = m["data_off"] + len(m["head"])
            fs = [self.ex.submit(_crc_and_write, self.fd, m["raw"][a:b], base + a)
                  for a, b in _crc_chunks(m["raw"], 4 * _THREADS)]
        self.jobs.append({"m": m, "fs": fs, "arr": keep})

    def _finish(self, job):
        """A member whose payload writes are done: its CRC-32 (the chunks' combined in order), and the
        writer's references to the payload dropped.  Raises what a failed write raised."""
        with self.lock:
            fs, m = job["fs"], job["m"]
            if fs is None:
                return
            if "crc" in m:
                fs[0].result()
            else:
                crc = zlib.crc32(m["head"])
                for f in fs:
                    c, n = f.result()
                    crc = crc32_combine(crc, c, n)
                m["crc"] = crc
            m["raw"] = None
            job["fs"] = job["arr"] = None

    def flush_async(self):
        """Start writing back what has been added so far (an fsync on a worker thread once those
        members' payload writes are done): close()'s own fsync then finds little left to do, and the
        arrays of those members are let go of."""
        snapshot = [j for j in self.jobs if j["fs"] is not None]

        def sync():
            for j in snapshot:
                for f in (j["fs"] or ()):
                    f.exception()       # (wait)
            for j in snapshot:
                self._finish(j)         # (a failed write raises here: close() reports it)
            os.fsync(self.fd)
        self.flushes.append(self.ex.submit(sync))

    def abort(self):
        if self.closed:
            return
        self.closed = True
        self.ex.shutdown(wait=True, cancel_futures=True)
        os.close(self.fd)
        try:
            os.unlink(self.tmp_path)
        except OSError:
            pass

    def close(self):
        ok = False
        fd, members = self.fd, self.members
        try:
            for f in self.flushes:
                f.result()
            for j in self.jobs:
                self._finish(j)
            cd_off = self.off
            # ---- headers and central directory
            cd = b""
            for m in members:
                # ZIP64 extra field of the local header: uncompressed, compressed size
                extra = (struct.pack("<HHQQ", 1, 16, m["usize"], m["csize"]) if m["z64"] else b"") + m["xtra"]
                flag = 0 if m["name"].isascii() else 0x800     # bit 11: the name is UTF-8
                lhdr = struct.pack("<IHHHHHIIIHH", 0x04034b50, 45 if m["z64"] else 20, flag, m["method"],
                                   _DOS_TIME, _DOS_DATE, m["crc"],
                                   0xFFFFFFFF if m["z64"] else m["csize"],
                                   0xFFFFFFFF if m["z64"] else m["usize"], len(m["name"]),
                                   len(extra)) + m["name"] + extra
                assert len(lhdr) == m["lhdr_len"]
                _pwrite_all(fd, memoryview(lhdr + m["head"]), m["offset"])
                fields = []
                if m["z64"]:
                    fields += [m["usize"], m["csize"]]
                if m["offset"] >= _Z64:
                    fields.append(m["offset"])
                extra = (struct.pack("<HH" + "Q" * len(fields), 1, 8 * len(fields), *fields) if fields else b"") \
                    + m["xtra"]
                cd += struct.pack("<IHHHHHHIIIHHHHHII", 0x02014b50, 45, 45 if fields else 20, flag,
                                  m["method"], _DOS_TIME, _DOS_DATE, m["crc"],
                                  0xFFFFFFFF if m["z64"] else m["csize"],
                                  0xFFFFFFFF if m["z64"] else m["usize"], len(m["name"]), len(extra),
                                  0, 0, 0, 0o600 << 16,
                                  0xFFFFFFFF if m["offset"] >= _Z64 else m["offset"]) \
                    + m["name"] + extra
            tail = b""
            if len(members) >= 0xFFFF or cd_off >= _Z64 or len(cd) >= _Z64:
                tail += struct.pack("<IQHHIIQQQQ", 0x06064b50, 44, 45, 45, 0, 0, len(members),
                                    len(members), len(cd), cd_off)
                tail += struct.pack("<IIQI", 0x07064b50, 0, cd_off + len(cd), 1)
            tail += struct.pack("<IHHHHIIH", 0x06054b50, 0, 0, min(len(members), 0xFFFF),
                                min(len(members), 0xFFFF),
                                0xFFFFFFFF if len(cd) >= _Z64 else len(cd),
                                0xFFFFFFFF if cd_off >= _Z64 else cd_off, 0)
            _pwrite_all(fd, memoryview(cd + tail), cd_off)
            os.fsync(fd)            # the bytes are on disk before the name points at them
            ok = True
        finally:
            self.closed = True
            self.ex.shutdown(wait=True, cancel_futures=True)
            os.close(fd)
            self.jobs = []
            if ok:
                os.replace(self.tmp_path, self.final_path)
            else:
                try:
                    os.unlink(self.tmp_path)
                except OSError:
                    pass
        return self.final_path


def save_npz(path, arrays, compress_small=True):
    w = NpzWriter(path, compress_small)
    try:
        for name, val in arrays.items():
            w.add(name, val)
    except BaseException:
        w.abort()
        raise
    return w.close()


def _pwrite_all(fd, view, offset):
    done = 0
    while done < len(view):
        done += os.pwrite(fd, view[done:done + (64 << 20)], offset + done)


def _crc_and_write(fd, view, offset):
    crc = 0
    for o in range(0, len(view), 16 << 20):           # (the piece is still in cache when written)
        piece = view[o:o + (16 << 20)]
        crc = zlib.crc32(piece, crc)
        _pwrite_all(fd, piece, offset + o)
    return crc, len(view)


def _member_bytes(data, zf, name):
    """One member of an in-memory ZIP, inflated in ONE zlib call (which releases the GIL: np.load's
    ZipExtFile inflates in small pieces under it) and CRC-checked.  None = leave it to np.load."""
    try:
        zi = zf.getinfo(name)
    except KeyError:
        return None
    if zi.compress_type not in (0, 8) or zi.flag_bits & 0x1:
        return None
    o = zi.header_offset
    if bytes(data[o:o + 4]) != b"PK\x03\x04":
        return None
    nlen, elen = struct.unpack_from("<HH", data, o + 26)
    start = o + 30 + nlen + elen
    raw = data[start:start + zi.compress_size]
    if len(raw) != zi.compress_size:
        raise zipfile.BadZipFile("truncated member {!r}".format(name))
    # (bufsize = the announced size: ONE output allocation -- with the default 16 KB the buffer is grown
    #  six times for a 0.8 MB member, each time under the GIL the inflate otherwise releases)
    out = zlib.decompress(raw, -15, max(1, zi.file_size)) if zi.compress_type == 8 else bytes(raw)
    if len(out) != zi.file_size or zlib.crc32(out) != zi.CRC:
        raise zipfile.BadZipFile("Bad CRC-32 for file {!r}".format(name))
    return out


def load_sample(path):
    """-> (sample dict "1".."24" -> int32 array, binsize) as written by `convert`
    (main.py:33-35, convert_tools.py:110-119).  The file is read whole and its two members are
    inflated by one zlib call each, so loader threads really run side by side (newref imports
    hundreds of samples); anything unusual about the archive goes through np.load."""
    with open(path, "rb") as fh:
        data = memoryview(fh.read())
    try:
        zf = zipfile.ZipFile(io.BytesIO(data))
        parts = [_member_bytes(data, zf, n) for n in ("sample.npy", "binsize.npy")]
    except (zipfile.BadZipFile, struct.error, zlib.error) as e:
        raise zipfile.BadZipFile("{}: {}".format(path, e))
    if any(p is None for p in parts):
        npz = np.load(path, encoding="latin1", allow_pickle=True)
        return npz["sample"].item(), int(npz["binsize"])
    sample, binsize = (np.lib.format.read_array(io.BytesIO(p), allow_pickle=True,
                                                pickle_kwargs={"encoding": "latin1"}) for p in parts)
    return sample.item(), int(binsize)


def save_sample(path, sample, binsize, quality=None):
    np.savez_compressed(path, binsize=binsize, sample=sample, quality=quality or {})


def _read_into(path, offset, view, want_crc=False):
    crc = 0
    with open(path, "rb", buffering=0) as fh:
        fh.seek(offset)
        done = 0
        while done < len(view):
            n = fh.readinto(view[done:done + (64 << 20)])
            if not n:
                raise IOError("short read in {}".format(path))
            if want_crc:
                crc = zlib.crc32(view[done:done + n], crc)
            done += n
    return crc


def _gf2_times(mat, vec):
    s, i = 0, 0
    while vec:
        if vec & 1:
            s ^= mat[i]
        vec >>= 1
        i += 1
    return s


def _libz_combine():
    """zlib's own crc32_combine (the Python module does not export it), if libz can be loaded."""
    try:
        import ctypes
        import ctypes.util
        z = ctypes.CDLL(ctypes.util.find_library("z") or "libz.so.1")
        fn = z.crc32_combine
        fn.argtypes = [ctypes.c_ulong, ctypes.c_ulong, ctypes.c_long]
        fn.restype = ctypes.c_ulong
        return fn if fn(zlib.crc32(b"ab"), zlib.crc32(b"cde"), 3) == zlib.crc32(b"abcde") else None
    except (OSError, AttributeError):
        return None


_LIBZ_COMBINE = _libz_combine()


def crc32_combine(crc1, crc2, len2):
    """CRC-32 of A + B from crc32(A), crc32(B) and len(B): libz's crc32_combine when loadable, else
    the same operator here."""
    if _LIBZ_COMBINE is not None and len2 > 0:
        return int(_LIBZ_COMBINE(crc1, crc2, len2)) & 0xFFFFFFFF
    return crc32_combine_py(crc1, crc2, len2)


def crc32_combine_py(crc1, crc2, len2):
    """zlib's crc32_combine restated: the operator that appends len2 zero bytes, by repeated squaring
    in GF(2)."""
    if len2 <= 0:
        return crc1
    odd = [0xEDB88320] + [1 << n for n in range(31)]          # one zero BIT
    even = [_gf2_times(odd, odd[n]) for n in range(32)]       # two
    odd = [_gf2_times(even, even[n]) for n in range(32)]      # four
    while True:
        even = [_gf2_times(odd, odd[n]) for n in range(32)]   # first pass: one zero BYTE
        if len2 & 1:
            crc1 = _gf2_times(even, crc1)
        len2 >>= 1
        if not len2:
            break
        odd = [_gf2_times(even, even[n]) for n in range(32)]
        if len2 & 1:
            crc1 = _gf2_times(odd, crc1)
        len2 >>= 1
        if not len2:
            break
    return crc1 ^ crc2


class Reference(dict):
    """The members of a reference .npz.  `deferred`: big members not read yet (load_reference(defer=...)),
    key -> read plan; ensure_loaded() brings a suffix's members in."""
    path = None
    check_crc = True

    def __init__(self, *a, **k):
        dict.__init__(self, *a, **k)
        self.deferred = {}


def _read_tail(path, offset, region_len, tail_bytes, dst, want_crc):
    buf = bytearray(region_len)
    _read_into(path, offset, memoryview(buf))
    _unstore_blocks(buf, tail_bytes, dst)
    return zlib.crc32(dst) if want_crc else 0


def _read_members(path, plans, check_crc, out):
    """plans: [(key, arr, data offset, CRC of the header bytes, CRC of the directory[, hybrid description])]
    -> out[key] = arr, read by worker threads, CRC-32 checked.  A member enters `out` only when ALL of the
    plans have been read and verified: another thread using `out` meanwhile never sees a half-filled table,
    and after an I/O or CRC failure nothing of this call is in it.  A hybrid member (NpzWriter._add_hybrid)
    is rebuilt from its fill value and the stored blocks of its tail: `offset` is then the start of the
    member's data."""
    done = {}
    with ThreadPoolExecutor(max_workers=_THREADS) as ex:
        pending = []
        for plan in plans:
            key, arr, off, head_crc, want = plan[:5]
            hyb = plan[5] if len(plan) > 5 else None
            if hyb is None:
                view = _raw_view(arr)
                futs = [(b - a, ex.submit(_read_into, path, off + a, view[a:b], check_crc))
                        for a, b in _crc_chunks(view, _THREADS)]
                pending.append((key, arr, futs, head_crc, want))
                continue
            items = arr.reshape(-1)
            n_fill = hyb["prefix_bytes"] // hyb["itemsize"]
            value = np.frombuffer(hyb["fill"], dtype=arr.dtype)[0]
            step = max(1 << 20, -(-n_fill // _THREADS))
            fills = [ex.submit(items[a:min(a + step, n_fill)].fill, value) for a in range(0, n_fill, step)]
            flat = items.view(np.uint8)
            tail = ex.submit(_read_tail, path, off + hyb["tail_off"], hyb["region_len"], hyb["tail_bytes"],
                             flat[hyb["prefix_bytes"]:], check_crc)
            pending.append((key, arr, (fills, tail, hyb), head_crc, want))
        yield                                   # (the caller's own reading runs beside the workers)
        for key, arr, futs, crc, want in pending:
            if isinstance(futs, tuple):
                fills, tail, hyb = futs
                for f in fills:
                    f.result()
                c = tail.result()
                if check_crc:
                    crc = crc32_combine(crc, _const_crc(hyb["fill"], hyb["prefix_bytes"]), hyb["prefix_bytes"])
                    crc = crc32_combine(crc, c, hyb["tail_bytes"])
            else:
                for n, f in futs:
                    c = f.result()
                    if check_crc:
                        crc = crc32_combine(crc, c, n)
            if check_crc and crc != want:
                raise IOError("{}: CRC-32 mismatch in member {}.npy (corrupted file)".format(path, key))
            done[key] = arr
    out.update(done)


def ensure_loaded(ref, suffix):
    """Read the deferred big members of `ref` whose key ends with `suffix` ("" = all of them).  The
    read plans leave `ref.deferred` only once their tables are in `ref`: after a failure the call can
    be repeated, and a later access raises instead of finding garbage."""
    deferred = getattr(ref, "deferred", None)
    if not deferred:
        return ref
    keys = [k for k in list(deferred) if k.endswith(suffix)]
    plans = []
    for k in keys:
        shape, dtype, fortran, off, head_crc, want = deferred[k][:6]
        plans.append((k, np.empty(shape, dtype=dtype, order="F" if fortran else "C"), off, head_crc, want)
                     + tuple(deferred[k][6:]))
    for _ in _read_members(ref.path, plans, ref.check_crc, ref):
        pass
    for k in keys:
        deferred.pop(k, None)
    return ref


def load_reference(path, defer=()):
    """All members of a reference .npz as a dict.  Stored (uncompressed) numeric members are read
    straight into their arrays by worker threads (sizes checked against the file, CRC-32 against
    the directory); the rest (small deflated or pickled members) goes through np.load.
    defer: key suffixes (".F", ".M"; "" = every one) whose BIG members are only planned, not read --
    a predict uses the autosomal tables and ONE gonosomal set, a third of the 2.5 GB at 15 kb is never
    needed (ensure_loaded(ref, suffix) reads a set; indexing a deferred key raises KeyError)."""
    out, direct = Reference(), []
    out.path = path
    check_crc = os.environ.get("WCX_NPZ_NO_CRC", "") in ("", "0")
    fsize = os.path.getsize(path)
    with zipfile.ZipFile(path) as zf, open(path, "rb") as fh:
        for info in zf.infolist():
            if not info.filename.endswith(".npy"):
                continue
            key = info.filename[:-4]
            hyb = _hybrid_info(info.extra) if info.compress_type == zipfile.ZIP_DEFLATED else None
            if hyb is not None:
                # [stored block: .npy header][constant rows, deflated][stored blocks: tail] (NpzWriter._add_hybrid)
                fh.seek(info.header_offset)
                lh = fh.read(30)
                nlen, elen = struct.unpack("<HH", lh[26:30])
                data_off = info.header_offset + 30 + nlen + elen
                fh.seek(data_off + 5)
                head = fh.read(hyb["head_len"])
                hfh = io.BytesIO(head)
                version = np.lib.format.read_magic(hfh)
                shape, fortran, dtype = (np.lib.format.read_array_header_1_0 if version == (1, 0)
                                         else np.lib.format.read_array_header_2_0)(hfh)
                nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
                hyb["region_len"] = info.compress_size - hyb["tail_off"]
                n_blocks = max(1, -(-hyb["tail_bytes"] // _SB_MAX))
                if (fortran or dtype.hasobject or dtype.itemsize != hyb["itemsize"]
                        or hyb["prefix_bytes"] + hyb["tail_bytes"] != nbytes
                        or hyb["head_len"] + nbytes != info.file_size
                        or hyb["region_len"] != hyb["tail_bytes"] + 5 * n_blocks
                        or data_off + info.compress_size > fsize):
                    raise IOError("{}: member {} is truncated or its sizes disagree".format(path, info.filename))
                head_crc = zlib.crc32(head)
                if any(key.endswith(sfx) for sfx in defer):
                    out.deferred[key] = (shape, dtype, False, data_off, head_crc, info.CRC, hyb)
                    continue
                direct.append((key, np.empty(shape, dtype=dtype), data_off, head_crc, info.CRC, hyb))
                continue
            if info.compress_type != zipfile.ZIP_STORED or info.file_size < _BIG:
                continue
            fh.seek(info.header_offset)
            lh = fh.read(30)
            nlen, elen = struct.unpack("<HH", lh[26:30])
            data_off = info.header_offset + 30 + nlen + elen
            fh.seek(data_off)
            version = np.lib.format.read_magic(fh)
            if version == (1, 0):
                shape, fortran, dtype = np.lib.format.read_array_header_1_0(fh)
            elif version == (2, 0):
                shape, fortran, dtype = np.lib.format.read_array_header_2_0(fh)
            else:
                continue
            if dtype.hasobject:
                continue
            nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
            head_len = fh.tell() - data_off
            if head_len + nbytes != info.file_size or data_off + info.file_size > fsize:
                raise IOError("{}: member {} is truncated or its sizes disagree ({} + {} bytes announced, "
                              "{} stored, file of {} bytes)".format(path, info.filename, head_len, nbytes,
                                                                    info.file_size, fsize))
            fh.seek(data_off)
            head_crc = zlib.crc32(fh.read(head_len))
            if any(key.endswith(sfx) for sfx in defer):
                out.deferred[key] = (shape, dtype, fortran, data_off + head_len, head_crc, info.CRC)
                continue
            arr = np.empty(shape, dtype=dtype, order="F" if fortran else "C")
            direct.append((key, arr, data_off + head_len, head_crc, info.CRC))
    out.check_crc = check_crc
    reader = _read_members(path, direct, check_crc, out)
    next(reader)                                # workers started
    planned = {p_[0] for p_ in direct}
    with np.load(path, encoding="latin1", allow_pickle=True) as npz:
        for k in npz.files:
            if k not in planned and k not in out.deferred:
                out[k] = npz[k]
    for _ in reader:                            # join + CRC
        pass
    return out
