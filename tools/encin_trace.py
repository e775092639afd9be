"""Perf study: phase timeline of encin.hip (-DEEND_ENCIN_TRACE variant library).  Numbers are shader cycles / 100."""
import ctypes, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
_lib = importlib.import_module("fs-eend_amd.lib"); ops = importlib.import_module("fs-eend_amd.ops")
t = importlib.import_module("test_hip_encin")
B, T, Fin = 64, 500, 345
src, lens, bn, w, bias, gamma, beta = t._case(B, T, Fin, 1, ragged=False)
Tp = ops.frames_pad(T)
o16 = torch.empty(B * Tp, 256, dtype=torch.float16, device="cuda")
for _ in range(3):
    ops.encoder_input(src, bn, w, bias, gamma, beta, None, o16, T, Tp, -1.0)
L = _lib.load()
tr = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
L.eend_debug_encin_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert L.eend_debug_encin_trace(tr.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
x = tr.view(256, 8, 8).cpu().double() / 100.0
for blk in (0, 200):
    print(f"block {blk}: [kernel start = 0] top, landed barrier, dma issued, frags built, mfma done, tile end")
    for ti in range(8):
        print("  tile", ti, " ".join(f"{v:8.2f}" for v in (x[blk, ti, :6] - x[blk, ti, 7]).tolist()))
d = x[:, :, 1:6] - x[:, :, 0:5]
print("mean phase durations:", [round(float(d[:, :, k].mean()), 2) for k in range(5)], " tile period", round(float((x[:, 1:, 0] - x[:, :-1, 0]).mean()), 2))
