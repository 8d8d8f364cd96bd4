# does hipExtStreamCreateWithCUMask work on this box, and how are the mask bits laid out?
import ctypes as C, torch, time
hip = C.CDLL("libamdhip64.so")
torch.zeros(1, device="cuda")
st = C.c_void_p()
mask = (C.c_uint32 * 8)(*([0xFFFFFFFF] * 8))
rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, mask)
print("create full mask rc", rc)
mask2 = (C.c_uint32 * 8)(0x0000FFFF, 0, 0, 0, 0, 0, 0, 0)
st2 = C.c_void_p()
rc = hip.hipExtStreamCreateWithCUMask(C.byref(st2), 8, mask2)
print("create 16-CU mask rc", rc)
x = torch.randn(1 << 26, device="cuda")
for name, s in (("default", None), ("full", st), ("16cu", st2)):
    if s is None:
        stream = torch.cuda.current_stream()
    else:
        stream = torch.cuda.ExternalStream(s.value)
    with torch.cuda.stream(stream):
        for _ in range(3): y = x * 2.0 + 1.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): y = x * 2.0 + 1.0
        torch.cuda.synchronize()
        print(name, "GB/s", 20 * 2 * 4 * (1 << 26) / (time.perf_counter() - t0) / 1e9)
