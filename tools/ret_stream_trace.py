"""Perf study: per-item phase timeline of ret_stream.hip's pass 2 (library built with -DEEND_RS_TRACE:
SRC=ret_stream tools/fs_variants.sh rstrace=-DEEND_RS_TRACE; run with EEND_HIP_LIB=.../libeend_hip_rstrace.so).  Shader-clock cycles / 100."""
import ctypes, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_lib = importlib.import_module("fs-eend_amd.lib"); ops = importlib.import_module("fs-eend_amd.ops")
dev = "cuda"
g = torch.Generator().manual_seed(0)
nseq, Tp, L, T = int(os.environ.get("NSEQ", 160)), 2048, 500, 2000
w = torch.randn(1024, 256, generator=g) / 16; b = torch.randn(1024, generator=g) * 0.1
ws = ops.retention_stream_pack(w.to(dev).contiguous()); b = b.to(dev)
x32 = torch.randn(nseq * Tp, 256, generator=g).to(dev)
x = x32.half(); xlo = (x32 - x.float()).half() if os.environ.get("XLO", "1") != "0" else None
o = torch.empty_like(x)
nc = (Tp + L - 1) // L
st = torch.empty(nseq * 4 * nc * 2 * 4096, dtype=torch.float16, device=dev)
cs = torch.empty(nseq * 4 * nc, device=dev); se = torch.empty(nseq * 4 * nc, device=dev)
for _ in range(3):
    ops.retention_stream(x, xlo, ws, b, o, st, cs, se, nseq, Tp, L, 1e-6, t_valid=T)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.retention_stream(x, xlo, ws, b, o, st, cs, se, nseq, Tp, L, 1e-6, t_valid=T)
e1.record(); torch.cuda.synchronize()
print(f"retention_stream nseq={nseq} Tp={Tp}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us per call (three launches)")
L_ = _lib.load()
if not hasattr(L_, "eend_debug_ret_stream_trace"):
    sys.exit(0)
tr = torch.zeros(256 * 8 * 32, dtype=torch.int64, device=dev)
L_.eend_debug_ret_stream_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert L_.eend_debug_ret_stream_trace(tr.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
t = tr.view(256, 8, 32).cpu().double()
NAMES = ["item top (rows requested)", "previous item's barrier", "xq fragments", "xlo fragments", "Q0H barrier", "Q1H barrier", "Q0L barrier", "Q1L barrier",
         "G0 barrier", "G1 barrier", "G done", "own K/V done", "other rows landed", "other fragments", "other K/V done", "K/V barrier", "block done",
         "q operands staged", "tile loop done", "cross term done", "", "", "", "", "K0 barrier", "K0 done", "K1 done", "V0 barrier", "V0 done", "V1 done"]
ORDER = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 24, 25, 26, 27, 28, 29, 11, 12, 13, 14, 15, 17, 18, 19, 16]
for up in (1, 0):
    sel = (t[:, :, 20] == up) & (t[:, :, 0] > 0)
    tt = t[sel][:, :30] / 100.0
    if tt.numel() == 0:
        continue
    print(f"{'upper' if up else 'lower'} items ({tt.shape[0]} samples): mean phase durations [cycles / 100]")
    prev = 0
    for k in ORDER[1:]:
        if (tt[:, k] == 0).all():
            continue
        d = tt[:, k] - tt[:, prev]
        print(f"  {NAMES[prev]:28s} -> {NAMES[k]:28s} {d.mean():8.2f}  (min {d.min():.2f}, max {d.max():.2f})")
        prev = k
    print(f"  whole item {(tt[:, 16] - tt[:, 0]).mean():.2f}")
