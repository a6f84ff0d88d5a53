"""NumPy dtype <-> gm_dtype codes (include/gmhip.h: enum gm_dtype)."""
import numpy as np

GM_BOOL, GM_U8, GM_I8, GM_U16, GM_I16, GM_U32, GM_I32, GM_U64, GM_I64, GM_F32, GM_F64 = range(11)

_TO_CODE = {
    np.dtype("bool"): GM_BOOL, np.dtype("u1"): GM_U8, np.dtype("i1"): GM_I8,
    np.dtype("u2"): GM_U16, np.dtype("i2"): GM_I16, np.dtype("u4"): GM_U32,
    np.dtype("i4"): GM_I32, np.dtype("u8"): GM_U64, np.dtype("i8"): GM_I64,
    np.dtype("f4"): GM_F32, np.dtype("f8"): GM_F64,
}
_FROM_CODE = {v: k for k, v in _TO_CODE.items()}


def code_of(dtype):
    dt = np.dtype(dtype)
    try:
        return _TO_CODE[dt]
    except KeyError:
        raise NotImplementedError("dtype %s is not supported by the HIP raster path" % dt)


def dtype_of(code):
    return _FROM_CODE[code]


def is_supported(dtype):
    return np.dtype(dtype) in _TO_CODE
