import torch, ctypes
p = torch.cuda.get_device_properties(0)
print({k: getattr(p, k) for k in dir(p) if not k.startswith('_') and 'shared' in k.lower()})
hip = ctypes.CDLL("libamdhip64.so")
v = ctypes.c_int()
for name, code in [("MaxSharedMemoryPerBlock", 8 + 66), ]:
    pass
for code in range(0, 120):
    r = hip.hipDeviceGetAttribute(ctypes.byref(v), code, 0)
    if r == 0 and v.value in (65536, 163840, 160 * 1024, 64 * 1024):
        print(code, v.value)
