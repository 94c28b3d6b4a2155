"""Small host-side value types mirroring the reference's Math/Core pieces the hot path uses.

BBox2i follows vw::math::BBox (src/vw/Math/BBox.tcc:82-197): half-open [min, max) integer boxes.
Exceptions mirror src/vw/Core/Exception.h:225-253; the C ABI's status codes map onto them exactly as
the C++ wrappers in vwlite do.
"""
import ctypes
import enum

from . import _lib


class VWException(Exception):
    """vw::Exception"""


class ArgumentErr(VWException):
    """vw::ArgumentErr"""


class NoImplErr(VWException):
    """vw::NoImplErr"""


class LogicErr(VWException):
    """vw::LogicErr"""


class CostFunctionType(enum.IntEnum):
    """vw::stereo::CostFunctionType, src/vw/Stereo/CostFunctions.h:143-149."""
    ABSOLUTE_DIFFERENCE = 0
    SQUARED_DIFFERENCE = 1
    CROSS_CORRELATION = 2
    CENSUS_TRANSFORM = 3
    TERNARY_CENSUS_TRANSFORM = 4


ABSOLUTE_DIFFERENCE = CostFunctionType.ABSOLUTE_DIFFERENCE
SQUARED_DIFFERENCE = CostFunctionType.SQUARED_DIFFERENCE
CROSS_CORRELATION = CostFunctionType.CROSS_CORRELATION

PATH_NONE, PATH_GENERIC_F64, PATH_SAD_U8, PATH_DOT_U8, PATH_EXACT_ORDER, PATH_SAD_U16, PATH_DOT_U16, PATH_REFUSED, PATH_CERTIFIED = 0, 1, 2, 3, 4, 5, 6, 7, 8
OPT_DEFER_EXACTNESS, OPT_DEVICE_COUNT, OPT_SAD_GROUPS, OPT_EXACT_SCRATCH_MB, OPT_TRACE, OPT_SGM_SWEEP = 1, 2, 3, 4, 5, 6
OPT_MGM_SWEEP, OPT_EXACT_SPLIT, OPT_HOST_RING_KB, OPT_HOST_RING_WRAPS, OPT_CERTIFY, OPT_CERT_PERMILLE, OPT_ZONE_SXC = 8, 9, 11, 12, 13, 14, 15      # vwgpu_option
OPT_CERT_F32, OPT_CERT_F64_PERMILLE, OPT_ZONE_TILE16, OPT_SGM_PATH_MODE = 16, 17, 18, 19
VALID_I32 = 0x7FFFFFFF


class BBox2i:
    """Half-open integer box [min, max); BBox2i(x, y, w, h) like the reference's 4-argument ctor."""

    def __init__(self, x=0, y=0, w=0, h=0):
        self.min = [int(x), int(y)]
        self.max = [int(x) + int(w), int(y) + int(h)]

    @classmethod
    def from_corners(cls, mn, mx):
        b = cls()
        b.min = [int(mn[0]), int(mn[1])]
        b.max = [int(mx[0]), int(mx[1])]
        return b

    def empty(self):
        return self.min[0] >= self.max[0] or self.min[1] >= self.max[1]

    def width(self):
        return 0 if self.empty() else self.max[0] - self.min[0]

    def height(self):
        return 0 if self.empty() else self.max[1] - self.min[1]

    def size(self):
        return (self.max[0] - self.min[0], self.max[1] - self.min[1])

    def __repr__(self):
        return "BBox2i(min=%s, max=%s)" % (tuple(self.min), tuple(self.max))


def bounding_box(img):
    """bounding_box(image): (0,0,cols,rows) for a (rows, cols[, ...]) array."""
    return BBox2i(0, 0, img.shape[1], img.shape[0])


_STATUS_EXC = {-1: ArgumentErr, -2: NoImplErr, -3: LogicErr, -4: LogicErr, -5: LogicErr}


class Context:
    """One engine context per (host thread x GPU): wraps vwgpu_create / vwgpu_destroy."""

    def __init__(self, device=0, stream=None):
        self._lib = _lib.load()
        h = ctypes.c_void_p()
        rc = self._lib.vwgpu_create(ctypes.byref(h), int(device))
        if rc != 0:
            raise LogicErr("vwgpu_create(device=%d) failed: %s (no GPU? there is no CPU fallback)"
                           % (device, self._lib.vwgpu_strerror(rc).decode()))
        self._h = h
        self.device = int(device)
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vwgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != 0:
            exc = _STATUS_EXC.get(rc, LogicErr)
            detail = self._lib.vwgpu_last_error(self._h).decode()
            raise exc(detail or self._lib.vwgpu_strerror(rc).decode())

    def set_stream(self, stream_ptr):
        """Enqueue on the given hipStream_t (int handle); 0/None = the legacy default stream (torch's default)."""
        self.check(self._lib.vwgpu_set_stream(self._h, ctypes.c_void_p(stream_ptr or None)))

    def reset_stream(self):
        self.check(self._lib.vwgpu_reset_stream(self._h))

    def synchronize(self):
        self.check(self._lib.vwgpu_synchronize(self._h))

    def trim(self):
        """vwgpu_trim: waits for the stream, gives every scratch arena back to the device; returns the bytes released."""
        n = ctypes.c_size_t(0)
        self.check(self._lib.vwgpu_trim(self._h, ctypes.byref(n)))
        return n.value

    def force_path(self, path):
        self.check(self._lib.vwgpu_force_path(self._h, int(path)))

    def last_path(self):
        return self._lib.vwgpu_last_path(self._h)

    def set_option(self, option, value):
        """vwgpu_set_option: e.g. (OPT_DEFER_EXACTNESS, 1) for pipelined callers that must not wait for the input-class flags."""
        self.check(self._lib.vwgpu_set_option(self._h, int(option), int(value)))

    def get_option(self, option):
        v = ctypes.c_int(0)
        self.check(self._lib.vwgpu_get_option(self._h, int(option), ctypes.byref(v)))
        return v.value

    def profile_enable(self, on=True):
        self.check(self._lib.vwgpu_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        self.check(self._lib.vwgpu_profile_reset(self._h))

    def profile_read(self, cap=4096):
        names = (ctypes.c_char_p * cap)()
        ms = (ctypes.c_float * cap)()
        n = self._lib.vwgpu_profile_read(self._h, names, ms, cap)
        if n < 0:
            self.check(n)
        return [(names[i].decode(), float(ms[i])) for i in range(n)]


_DEFAULT = {}


def default_context(device=0):
    ctx = _DEFAULT.get(device)
    if ctx is None:
        ctx = _DEFAULT[device] = Context(device)
    return ctx
