"""hipMemsetAsync nodes inside a captured HIP graph: on ROCm 7.2 / gfx950 the first memset of this three-memset sequence
stops taking effect from the second replay on (prints `hdr zero: False`).  libmacr_hip therefore fills with kernels."""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)
buf = torch.zeros(2 << 20, dtype=torch.uint8, device=dev)
HDR, TAU, U, MAX, MAXB = 42240, 42240, 2090, 50688, 535040
def fills():
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    base = buf.data_ptr()
    r = [hip.hipMemsetAsync(ctypes.c_void_p(base), ctypes.c_int(0), ctypes.c_size_t(HDR), st),
         hip.hipMemsetAsync(ctypes.c_void_p(base + MAX), ctypes.c_int(0xff), ctypes.c_size_t(MAXB), st),
         hip.hipMemsetD32Async(ctypes.c_void_p(base + TAU), ctypes.c_int(-8388608), ctypes.c_size_t(U), st)]
    return r
def check(tag):
    torch.cuda.synchronize()
    h = buf[:HDR]; t = buf[TAU:TAU + 4 * U].view(torch.float32); m = buf[MAX:MAX + MAXB]
    print(tag, "hdr zero:", int(h.sum()) == 0, "| tau -inf:", int(torch.isinf(t).sum()), "of", U, t[:3].tolist(), "| maxima ff:", int((m == 255).sum()) == MAXB, "| after-maxima untouched:", int(buf[MAX + MAXB:MAX + MAXB + 64].sum()))
buf.fill_(3); print(fills()); check("eager ")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    print(fills())
for k in range(3):
    buf.fill_(3); g.replay(); check("replay%d" % k)
