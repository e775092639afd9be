"""Same-box timing of encoder_input (encin.hip) against gather_bn_cast_pad + linear_res_ln at the FS model.test shape."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
ops = importlib.import_module("fs-eend_amd.ops")
t = importlib.import_module("test_hip_encin")
B, T, Fin = 64, 500, 345
src, lens, bn, w, bias, gamma, beta = t._case(B, T, Fin, 1, ragged=False)
Tp = ops.frames_pad(T)
o16 = torch.empty(B * Tp, 256, dtype=torch.float16, device="cuda"); p16 = torch.empty_like(o16)
x16 = torch.zeros(B * Tp, w.shape[1], dtype=torch.float16, device="cuda")
fns = {"encoder_input": lambda: ops.encoder_input(src, bn, w, bias, gamma, beta, None, o16, T, Tp, -1.0),
       "gather + linear_res_ln": lambda: (ops.gather_bn_cast_pad(src, bn, x16, T, Tp, -1.0, True), ops.linear_res_ln(x16, w, bias, None, gamma, beta, None, p16))}
res = {k: [] for k in fns}
for _ in range(5):
    for k, fn in fns.items():
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / 20 * 1e3)
print("max |diff|", (o16.float() - p16.float()).abs().max().item())
for k, v in res.items():
    print(f"{k}: min {min(v):.1f} us  median {sorted(v)[2]:.1f} us")
