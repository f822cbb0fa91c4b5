"""ASDG1: the container tools/dump_reference_golden.cpp writes on the reference side and tests/test_reference_golden.py reads.

    "ASDG" | u32 version = 1 | u32 n_arrays | n_arrays x { u16 name_len | name | u8 dtype | u8 ndim | u32 dims[ndim] | data }

dtype: 0 = u8, 1 = i32, 2 = f32, 3 = f64; little endian, C order.  Arrays of one front-end dump (L = level 0..nlevels-1):
  params        i32[5]   nfeatures, nlevels, iniThFAST, minThFAST, scaleFactor * 1000
  image         u8[h][w] the input frame
  pyr_L         u8[][]   mvImagePyramid[L] WITHOUT the 19 px border (ORBextractor.cc:1251-1276)
  blur_L        u8[][]   GaussianBlur(pyr_L, 7x7, sigma 2, BORDER_REFLECT_101) (:1226-1227)
  raw_L         f32[n][3] FAST corners of level L in vToDistributeKeys order: x, y (relative to minBorder), response (:813-876)
  keypoints     f32[n][6] ExtractDesc's output keypoints: x, y, size, angle, response, octave (:1234-1245)
  atan_in       f32[m][2] (y, x) samples;  atan_out f32[m] = cv::fastAtan2(y, x)
"""
import struct

import numpy as np

_DT = {0: np.uint8, 1: np.int32, 2: np.float32, 3: np.float64}
_CODE = {np.dtype(v): k for k, v in _DT.items()}


def read(path):
    buf = open(path, "rb").read()
    if buf[:4] != b"ASDG":
        raise ValueError(f"{path}: not an ASDG file")
    version, n = struct.unpack_from("<II", buf, 4)
    if version != 1:
        raise ValueError(f"{path}: version {version}")
    off, out = 12, {}
    for _ in range(n):
        (ln,) = struct.unpack_from("<H", buf, off); off += 2
        name = buf[off:off + ln].decode(); off += ln
        code, nd = struct.unpack_from("<BB", buf, off); off += 2
        dims = struct.unpack_from(f"<{nd}I", buf, off); off += 4 * nd
        dt = np.dtype(_DT[code])
        cnt = int(np.prod(dims)) if nd else 1
        out[name] = np.frombuffer(buf, dt, cnt, off).reshape(dims).copy()
        off += cnt * dt.itemsize
    return out


def write(path, arrays):
    with open(path, "wb") as f:
        f.write(b"ASDG" + struct.pack("<II", 1, len(arrays)))
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            nb = name.encode()
            f.write(struct.pack("<H", len(nb)) + nb + struct.pack("<BB", _CODE[a.dtype], a.ndim))
            f.write(struct.pack(f"<{a.ndim}I", *a.shape))
            f.write(a.tobytes())
