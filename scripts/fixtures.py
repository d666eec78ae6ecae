"""Readers for the reference repository's own example images (only usable where
/root/reference exists, i.e. in the build container -- never on the GPU box).
Used by scripts/make_golden.py to produce tests/golden/*.npz."""
import lzma
import struct

import numpy as np

REF = "/root/reference"


def read_rds_int_matrix(path):
    """Minimal reader for an XZ-compressed R serialization (version 2/3, XDR) holding one
    INTSXP matrix with a dim attribute -- enough for CornerDetectionHarris/inst/extdata/building.rds."""
    raw = lzma.open(path).read()
    assert raw[:2] == b"X\n", raw[:8]
    off = 2
    version, = struct.unpack(">i", raw[off:off + 4]); off += 12  # version, writer, min reader
    if version == 3:
        n, = struct.unpack(">i", raw[off:off + 4]); off += 4 + n  # native encoding string
    flags, = struct.unpack(">I", raw[off:off + 4]); off += 4
    sxp = flags & 0xFF
    assert sxp == 13, f"expected INTSXP, got {sxp}"
    length, = struct.unpack(">i", raw[off:off + 4]); off += 4
    data = np.frombuffer(raw, dtype=">i4", count=length, offset=off).astype(np.int32)
    off += 4 * length
    # attributes pairlist: find the dim INTSXP of length 2 by scanning for it
    rest = raw[off:]
    idx = rest.find(b"dim")
    assert idx >= 0
    p = idx + 3
    f2, l2 = struct.unpack(">Ii", rest[p:p + 8])
    assert f2 & 0xFF == 13 and l2 == 2
    dim = struct.unpack(">ii", rest[p + 8:p + 16])
    return data.reshape(dim[1], dim[0])  # R column-major (nrow, ncol) -> C array [col][row]
