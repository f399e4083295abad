import sys, torch, ctypes as C
sys.path.insert(0, '.')
from lowlightenvironmentvideoobjectdetection_amd._lib import lib
dev = torch.device('cuda:0')
L = lib()
for rows, c in ((942080, 256), (942080 * 2, 64), (235520, 512)):
    a = torch.randn(rows, c, device=dev).half(); b = torch.randn(rows, c, device=dev).half(); y = torch.empty_like(a)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run():
        L.llv_add_f16(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(y.data_ptr()), C.c_int64(rows), c, c, c, c, st)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f'add_f16 {rows}x{c}: {ms*1e3:.1f} us, {3 * rows * c * 2 / ms / 1e9:.2f} TB/s (2 reads + 1 write)')
    e0.record()
    for _ in range(20): y.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f'torch copy  {rows}x{c}: {ms*1e3:.1f} us, {2 * rows * c * 2 / ms / 1e9:.2f} TB/s (1 read + 1 write)')
