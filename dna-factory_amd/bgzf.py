"""BGZF writer (the sink right after the hot path; the reference uses Bio.bgzf.BgzfWriter, pop_factory.py:403).

BGZF = a series of gzip members, each holding at most 64 KiB of input, each with a `BC` extra field giving the
member's size, ended by an empty 28-byte member (SAM/BAM specification, section 4.1).  Blocks are compressed
independently, so they are compressed on a thread pool (zlib releases the GIL) and written in order.
biopython's compressed bytes are not reproduced (parity is defined on the decompressed stream);
any gzip reader, `bgzip -d` and tabix-style indexing accept the output.
"""
import io
import struct
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


class BgzfWriter:
    """write(bytes | str | buffer) ... close().  `threads` block compressors; output order = input order."""

    def __init__(self, filename=None, mode="wb", fileobj=None, compresslevel=6, threads=4, max_pending=256):
        if fileobj is None:
            if "w" not in mode and "a" not in mode:
                raise ValueError("BgzfWriter is write-only")
            fileobj = open(filename, "ab" if "a" in mode else "wb")
        self._f = fileobj
        self._level = compresslevel
        self._buf = bytearray()
        self._pool = ThreadPoolExecutor(max_workers=max(1, threads))
        self._pending = []
        self._max_pending = max_pending
        self.bytes_in = 0
        self.bytes_out = 0

    def _drain(self, keep):
        while len(self._pending) > keep:
            blk = self._pending.pop(0).result()
            self._f.write(blk)
            self.bytes_out += len(blk)

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
        """Append BGZF members that were produced elsewhere (the GPU encoder), after everything written so far."""
        self.flush()
        mv = memoryview(blob).cast("B")
        self._f.write(mv)
        self.bytes_out += len(mv)

    def flush(self):
        if self._buf:
            self._pending.append(self._pool.submit(compress_block, bytes(self._buf), self._level))
            self._buf = bytearray()
        self._drain(0)
        self._f.flush()

    def close_pool(self):
        """Flush without writing the EOF block and release the worker threads (shard files that get concatenated)."""
        self.flush()
        self._f = None
        self._pool.shutdown()

    def close(self):
        if self._f is None:
            return
        self.flush()
        self._f.write(EOF_BLOCK)
        self.bytes_out += len(EOF_BLOCK)
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
