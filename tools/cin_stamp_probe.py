import sys, ctypes, torch
sys.path.insert(0, '.')
import dir_amd
from dir_amd import ops, _lib
lib = dir_amd.load_library()
B, m, D = 65536, 26, 16
g = torch.Generator(device='cuda').manual_seed(1)
x0 = torch.randn((B, m, D), generator=g, device='cuda') * 0.25
x1 = torch.randn((B, 128, D), generator=g, device='cuda') * 0.25
W2 = torch.randn((128, 128 * m), generator=g, device='cuda') * 0.02
buf = (ctypes.c_uint64 * 8)()
for _ in range(2):
    ops.cin_layer(x0, x1, W2)
lib.dir_debug_cin_stamps(buf)
for _ in range(3):
    ops.cin_layer(x0, x1, W2)
lib.dir_debug_cin_stamps(buf)
a, b, cnt, pro, epi, waves = [buf[i] for i in range(6)]
print("chunks/wave %.1f  mfma-span cyc/chunk %.0f  post(bulk+barrier) cyc/chunk %.0f  prologue cyc/wave %.0f  epilogue cyc/wave %.0f" % (cnt / waves, a / cnt, b / cnt, pro / waves, epi / waves))
print("ideal MFMA cycles per chunk: %d" % (416 * 64))
