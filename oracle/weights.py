"""Reader of the SVADW001 container (tools/export_weights.py) for tests."""
import struct

import numpy as np


def read_container(blob: bytes) -> dict:
    assert blob[:8] == b"SVADW001"
    (n,) = struct.unpack_from("<I", blob, 8)
    out = {}
    for i in range(n):
        name, ndim, d0, d1, d2, d3, off, cnt = struct.unpack_from("<64sI4IQQ", blob, 80 + 100 * i)
        shape = (d0, d1, d2, d3)[:ndim]
        out[name.rstrip(b"\0").decode()] = np.frombuffer(blob, "<f4", cnt, off).reshape(shape)
    return out
