import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "avail"
import torch
if mode == "avail":
    print("torch avail", torch.cuda.is_available())
elif mode == "count":
    print("torch count", torch.cuda.device_count())
elif mode == "none":
    pass
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "nested_hashing_psi_amd", "libpiehip.so"))
hip = ctypes.CDLL("libamdhip64.so.7")
n = ctypes.c_int(-1)
hip.hipGetErrorString.restype = ctypes.c_char_p
rc = hip.hipGetDeviceCount(ctypes.byref(n)); print("count rc", rc, hip.hipGetErrorString(rc), n.value)
rc = hip.hipInit(0); print("init rc", rc)
rc = hip.hipGetDeviceCount(ctypes.byref(n)); print("count2 rc", rc, n.value)
for line in open('/proc/self/maps'):
    if ('amdhip' in line or 'hsa-runtime' in line) and 'r-xp' in line: print(line.strip())
from nested_hashing_psi_amd import pie
try:
    cc = pie.PieContext(1024, 2, 65537)
    print("ctx ok")
except Exception as e:
    print("ERR", e)
print("torch avail after", torch.cuda.is_available())
