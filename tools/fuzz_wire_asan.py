import ctypes as C, sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from pir_amd import capi
import test_host_logic as T
lib = C.CDLL(os.environ.get('WIRE_ASAN_LIB', '/tmp/libwire_asan.so'))
lib.pirgpu_wire_validate_request.argtypes = [C.POINTER(capi.Params), C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_uint32)]
lib.pirgpu_wire_validate_request.restype = C.c_int
req, p = T._wire_fixture()
assert T._validate(lib, p, req) == (0, 1)
rng = np.random.default_rng(5)
allowed = {0, 3, 12}
for cut in list(range(0, 200)) + [1000, len(req)//2, len(req)-9, len(req)-1]:
    assert T._validate(lib, p, req[:cut])[0] in (3, 12)
data = bytearray(req)
for it in range(3000):
    m = bytearray(data)
    for _ in range(int(rng.integers(1, 6))):
        pos = int(rng.integers(0, min(len(m), 4096 if it % 2 else len(m))))
        m[pos] ^= 1 << int(rng.integers(0, 8))
    assert T._validate(lib, p, bytes(m))[0] in allowed
for _ in range(500):
    g = rng.integers(0, 256, int(rng.integers(1, 8192)), dtype=np.uint8).tobytes()
    assert T._validate(lib, p, g)[0] in allowed
print("wire fuzz under ASan/UBSan OK")
