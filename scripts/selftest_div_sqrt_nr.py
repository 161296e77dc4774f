import ctypes as C, torch, sys
sys.path.insert(0, '/root/repo')
from kajiya_amd import lib
L = lib.load()
c = torch.zeros(4, dtype=torch.int64, device='cuda')
f = L.kj_selftest_div_sqrt_nr; f.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]; f.restype = C.c_int
n = 1 << 28
import os
seed = int(os.environ.get("SEED", "12345"))
assert f(n, seed, c.data_ptr(), None) == 0
torch.cuda.synchronize()
print("of", n, "pairs: quotients differing", int(c[0]), "roots differing", int(c[1]), "| off by more than an ulp:", int(c[2]), int(c[3]))
