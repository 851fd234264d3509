"""Legacy `index.faiss` files: read/write of the flat-index container that
`faiss.write_index` / `faiss.read_index` produce at `core.py:987` / `core.py:1057`, so an index saved by the
reference loads into the HBM-resident index and vice versa.

`faiss` is an un-vendored, un-pinned dependency that is absent from this image; the layout below restates its
published serialisation of `IndexFlat` (impl/index_write.cpp) from the format's documentation and is therefore
UNPINNED here (only the round trip through this module is tested):

    uint32  fourcc            "IxFI" (inner product) or "IxF2" (L2), little endian
    int32   d
    int64   ntotal
    int64   dummy, dummy      (1 << 20 each)
    uint8   is_trained
    int32   metric_type       0 = METRIC_INNER_PRODUCT, 1 = METRIC_L2
    uint64  count             number of float32 values = ntotal * d
    float32 xb[count]         rows, row-major
"""
import struct

import numpy as np

FOURCC_IP = struct.unpack("<I", b"IxFI")[0]
FOURCC_L2 = struct.unpack("<I", b"IxF2")[0]
_HEADER = struct.Struct("<IiqqqBiQ")      # packed, no padding: 4+4+8+8+8+1+4+8 = 45 bytes


def write_flat_index(path, vectors, metric="ip"):
    v = np.ascontiguousarray(vectors, dtype=np.float32)
    if v.ndim != 2:
        raise ValueError("vectors must be [n,d]")
    n, d = v.shape
    with open(path, "wb") as f:
        f.write(_HEADER.pack(FOURCC_IP if metric == "ip" else FOURCC_L2, d, n, 1 << 20, 1 << 20, 1,
                             0 if metric == "ip" else 1, n * d))
        f.write(v.tobytes())


def read_flat_index(path):
    """-> (vectors float32 [n,d], metric "ip" | "l2").  Raises ValueError for anything but a flat index."""
    with open(path, "rb") as f:
        head = f.read(_HEADER.size)
        if len(head) < _HEADER.size:
            raise ValueError(f"{path}: truncated header")
        fourcc, d, n, _, _, trained, metric, count = _HEADER.unpack(head)
        if fourcc not in (FOURCC_IP, FOURCC_L2):
            raise ValueError(f"{path}: not a flat FAISS index (fourcc {fourcc:#x}); only IndexFlatIP / IndexFlatL2 files "
                             "are supported (the reference coerces every configured type to IndexFlatIP, core.py:1205-1219)")
        if d <= 0 or n < 0 or count != n * d:
            raise ValueError(f"{path}: inconsistent header d={d} ntotal={n} count={count}")
        data = np.fromfile(f, dtype=np.float32, count=count)
        if data.size != count:
            raise ValueError(f"{path}: truncated payload ({data.size} of {count} floats)")
    return data.reshape(n, d), ("ip" if metric == 0 else "l2")
