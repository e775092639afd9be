"""Perf study: per-tile phase timeline of the decoder layer-head kernel (library built with -DEEND_SPK_TRACE:
SRC=spk_stream tools/fs_variants.sh spktrace=-DEEND_SPK_TRACE; run with EEND_HIP_LIB=.../libeend_hip_spktrace.so)."""
import ctypes, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_lib = importlib.import_module("fs-eend_amd.lib"); ops = importlib.import_module("fs-eend_amd.ops")
dev = "cuda"
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
R32 = os.environ.get("SPK_TRACE_R32", "0") == "1"      # the LS decoder form: f32 residual stream, 16 x 10 slots x Tp = 2048
B, C, Tp = (16, 10, 2048) if R32 else (64, 6, 512)
M = B * C * Tp
a = r(M, 256).half(); res = r(M, 256) if R32 else r(M, 256).half()
wo = r(256, 256, sc=1 / 16).half(); win = r(768, 256, sc=1 / 8).half()
bo = r(256, sc=0.1); g1 = 1 + r(256, sc=0.1); be1 = r(256, sc=0.1); bin_ = r(768, sc=0.3)
ws = ops.spk_stream_pack(wo, win)
x1 = torch.empty_like(res); o1 = torch.empty_like(a)
for _ in range(3):
    if R32:
        ops.attnout_spk_stream_res32(a, ws, bo, res, g1, be1, 1e-5, x1, bin_, o1, B, C, Tp)
    else:
        ops.attnout_spk_stream(a, ws, bo, res, g1, be1, 1e-5, x1, bin_, o1, B, C, Tp)
print(f"spk_stream trace: B={B} C={C} Tp={Tp} {'f32 residual (R32)' if R32 else 'f16 residual'}")
L = _lib.load()
tr = torch.zeros(256 * 4 * 12, dtype=torch.int64, device=dev)
L.eend_debug_spk_stream_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert L.eend_debug_spk_stream_trace(tr.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
t = tr.view(256, 4, 12).cpu().double() / 100.0            # s_memtime ticks (shader clock on this part) -> units of 100 cycles
NAMES = ["tile start", "Wo items done", "LN11 + x1 stores done"] + [f"head {h} {w}" for h in range(4) for w in ("items done", "attention + O stores done")]
for blk in (0, 255):
    print(f"block {blk}: [100 cycles since the block's first stamp]")
    for ti in range(4):
        print(f"  tile {ti}: " + " ".join(f"{v:6.2f}" for v in (t[blk, ti, :11] - t[blk, 0, 0]).tolist()))
dur = t[:, :, 1:11] - t[:, :, 0:10]
print("mean phase durations over all CUs and tiles [100 cycles]:")
for k in range(10):
    print(f"  {NAMES[k]:34s} -> {NAMES[k + 1]:34s} {dur[:, :, k].mean():6.2f}  (min {dur[:, :, k].min():.2f}, max {dur[:, :, k].max():.2f})")
print(f"  tile end -> next tile start {(t[:, 1:, 0] - t[:, :-1, 10]).mean():.2f}; tile period {(t[:, 1:, 0] - t[:, :-1, 0]).mean():.2f}; block total {(t[:, 3, 10] - t[:, 0, 0]).mean():.2f}")
