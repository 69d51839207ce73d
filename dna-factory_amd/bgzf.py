"""BGZF writer (the sink right after the hot path; the reference uses Bio.bgzf.BgzfWriter, pop_factory.py:403).

BGZF = a series of gzip members, each holding at most 64 KiB of input, each with a `BC` extra field giving the
member's size, ended by an empty 28-byte member (SAM/BAM specification, section 4.1).  Blocks are compressed
independently, so they are compressed on a thread pool (zlib releases the GIL) and written in order.
biopython's compressed bytes are not reproduced (parity is defined on the decompressed stream);
any gzip reader, `bgzip -d` and tabix-style indexing accept the output.
"""
import io
import os
import struct
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

BLOCK_INPUT = 65280  # bytes of input per block (same choice as biopython / htslib: leaves room for incompressible data)
EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def compress_block(data, level):
    """One BGZF member for `data` (<= 65280 bytes)."""
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = c.compress(data) + c.flush()
    if len(body) + 26 > 65536:  # incompressible input: store it (raw deflate level 0 adds 5 bytes per 65535)
        c = zlib.compressobj(0, zlib.DEFLATED, -15)
        body = c.compress(data) + c.flush()
    bsize = len(body) + 25  # total member size - 1
    head = struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, 0x42, 0x43, 2, bsize)
    return head + body + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


_io_seconds = [0.0, 0]  # DNAF_TRACE_TILES: seconds spent inside pwrite by the I/O threads, bytes written


def _pwrite_all(fd, mv, offset):
    t0 = time.time()
    total = len(mv)
    while len(mv):
        n = os.pwrite(fd, mv, offset)
        mv = mv[n:]
        offset += n
    _io_seconds[0] += time.time() - t0
    _io_seconds[1] += total


class BgzfWriter:
    """write(bytes | str | buffer) ... close().  `threads` block compressors; output order = input order.

    A file opened by name is written with positional writes (os.pwrite at offsets the writer keeps itself), so that
    members produced elsewhere -- write_members, the GPU encoder's output -- can be handed to `io_threads` I/O threads
    (DNAF_IO_THREADS overrides): a pwrite from the GPU's pinned staging buffers runs at 9 GB/s where one from pageable
    memory runs at 12, and a writer that also waits for its hand-overs wrote 7.6 GB/s in all; two or three calls in flight
    together reach the 10-11 GB/s the box's page cache takes (profiles/r03_concat.txt).  With a `fileobj` everything is
    written in sequence."""

    def __init__(self, filename=None, mode="wb", fileobj=None, compresslevel=6, threads=4, max_pending=256, io_threads=2):
        self._fd = None
        self._pos = 0
        if fileobj is None:
            if "w" not in mode and "a" not in mode:
                raise ValueError("BgzfWriter is write-only")
            self._fd = os.open(filename, os.O_WRONLY | os.O_CREAT | (0 if "a" in mode else os.O_TRUNC), 0o644)
            if "a" in mode:
                self._pos = os.fstat(self._fd).st_size
        self._f = fileobj
        self._level = compresslevel
        self._buf = bytearray()
        self._pool = ThreadPoolExecutor(max_workers=max(1, threads))
        io_threads = int(os.environ.get("DNAF_IO_THREADS", io_threads))
        self._io = ThreadPoolExecutor(max_workers=max(1, io_threads)) if self._fd is not None else None
        self._io_pending = []
        self._pending = []
        self._max_pending = max_pending
        self.bytes_in = 0
        self.bytes_out = 0

    def _put(self, blk):
        if self._fd is not None:
            _pwrite_all(self._fd, memoryview(blk), self._pos)
            self._pos += len(blk)
        else:
            self._f.write(blk)
        self.bytes_out += len(blk)

    def _drain(self, keep):
        while len(self._pending) > keep:
            self._put(self._pending.pop(0).result())

    def write(self, data):
        if isinstance(data, str):
            data = data.encode("ascii")
        mv = memoryview(data).cast("B")
        self.bytes_in += len(mv)
        pos = 0
        if self._buf:
            take = min(BLOCK_INPUT - len(self._buf), len(mv))
            self._buf += mv[:take]
            pos = take
            if len(self._buf) == BLOCK_INPUT:
                self._pending.append(self._pool.submit(compress_block, bytes(self._buf), self._level))
                self._buf = bytearray()
        while len(mv) - pos >= BLOCK_INPUT:
            self._pending.append(self._pool.submit(compress_block, bytes(mv[pos: pos + BLOCK_INPUT]), self._level))
            pos += BLOCK_INPUT
            if len(self._pending) > self._max_pending:
                self._drain(self._max_pending // 2)
        if pos < len(mv):
            self._buf += mv[pos:]

    def write_members(self, blob):
        """Append BGZF members that were produced elsewhere (the GPU encoder), after everything written so far.
        Returns a future that is done when `blob` has been written (the caller may not reuse the buffer before), or
        None when the write has already happened (fileobj mode)."""
        self._flush_blocks()
        mv = memoryview(blob).cast("B")
        self.bytes_out += len(mv)
        if self._fd is None:
            self._f.write(mv)
            return None
        off = self._pos
        self._pos += len(mv)
        self._io_pending = [f for f in self._io_pending if not f.done() or f.result() is not None]  # (result() re-raises a failed write)
        fut = self._io.submit(_pwrite_all, self._fd, mv, off)
        self._io_pending.append(fut)
        return fut

    def _flush_blocks(self):
        if self._buf:
            self._pending.append(self._pool.submit(compress_block, bytes(self._buf), self._level))
            self._buf = bytearray()
        self._drain(0)

    def flush(self):
        self._flush_blocks()
        for f in self._io_pending:
            f.result()
        self._io_pending = []
        if self._f is not None:
            self._f.flush()

    def close(self):
        if self._fd is None and self._f is None:
            return
        self.flush()
        self._put(EOF_BLOCK)
        if os.environ.get("DNAF_TRACE_TILES") and _io_seconds[1]:
            print("writer: %.2f s inside pwrite for %.1f GB (%.1f GB/s per call)" % (_io_seconds[0], _io_seconds[1] / 1e9, _io_seconds[1] / 1e9 / _io_seconds[0]), flush=True)
        if self._fd is not None:
            os.close(self._fd)
            self._fd = None
            self._io.shutdown()
        else:
            self._f.close()
            self._f = None
        self._pool.shutdown()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def read_all(path):
    """Decompress a whole BGZF / multi-member gzip file (test helper and small files only)."""
    import gzip

    with gzip.open(path, "rb") as f:
        return f.read()


def block_sizes(path):
    """Sizes of the BGZF members of a file, checking each header's BC field (used by tests)."""
    sizes = []
    with open(path, "rb") as f:
        while True:
            head = f.read(18)
            if not head:
                break
            magic, xlen, si1, si2, slen, bsize = struct.unpack("<4sxxxxxxHBBHH", head)
            if magic != b"\x1f\x8b\x08\x04" or (si1, si2, slen) != (0x42, 0x43, 2) or xlen != 6:
                raise ValueError("not a BGZF member at offset %d" % (f.tell() - 18))
            f.seek(bsize + 1 - 18, io.SEEK_CUR)
            sizes.append(bsize + 1)
    return sizes
