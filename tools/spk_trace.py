"""Perf study: per-tile phase timeline of the speaker-axis fused kernel (library built with -DEEND_SPK_TRACE:
tools/ab_variants.sh build spktrace=-DEEND_SPK_TRACE; run with EEND_HIP_LIB=.../libeend_hip_spktrace.so)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fs_eend_amd import lib as _lib, ops
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
B, C, Tp, T, H = 64, 6, 512, 500, 4
M = B * C * Tp
x = torch.randn(M, 256, generator=g).to(dev).half()
w = (torch.randn(768, 256, generator=g) * 0.06).to(dev).half()
bias = (torch.randn(768, generator=g) * 0.2).to(dev)
o = torch.empty(M, 256, dtype=torch.float16, device=dev)
NAMES = ["tile start", "X rows gathered into LDS", "X fragments in registers", "head 0: q slice done", "head 0: k slice done",
         "head 0: v slice done", "head 0: attention done", "head 1 done", "head 2 done", "head 3 done (tile end)"]
L = _lib.load()
for _ in range(3):
    ops.spk_qkv_attn(x, w, bias, o, B, C, Tp, H, t_valid=T)
tr = torch.zeros(256 * 16 * 12, dtype=torch.int64, device=dev)
L.eend_debug_spk_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert L.eend_debug_spk_trace(tr.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
t = tr.view(256, 16, 12).cpu().double() / 2000.0          # s_memtime counts shader clocks: ~us at ~2.0 GHz (check against the kernel time)
nt = 6
d = t[:, :nt, :10]
b0 = d[0, 0, 0]
print("block 0, tile phases [us since its first stamp]:")
for ti in range(nt):
    print(f"  tile {ti}: " + "  ".join(f"{(d[0, ti, k] - b0):7.2f}" for k in range(10)))
dur = d[:, :, 1:] - d[:, :, :-1]
print("mean phase durations over all CUs and tiles [us]:")
for k in range(9):
    print(f"  {NAMES[k]:28s} -> {NAMES[k + 1]:28s} {dur[:, :, k].mean():7.2f}  (min {dur[:, :, k].min():.2f}, max {dur[:, :, k].max():.2f})")
print(f"  tile end -> next tile start: {(d[:, 1:, 0] - d[:, :-1, 9]).mean():.2f}")
print(f"  tile period: {(d[:, 1:, 0] - d[:, :-1, 0]).mean():.2f}; block total {(d[:, nt - 1, 9] - d[:, 0, 0]).mean():.2f}")
