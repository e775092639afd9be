"""Same-box timing of attnout_spk_stream against the two launches it replaces (linear_res16_ln + linear + spk_attn) at the
FS model.test decoder shape (B=64, C=6, Tp=512).  Usage: python tools/ab_spk_stream.py [reps]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


import torch

ops = importlib.import_module("fs-eend_amd.ops")


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    B, C, Tp = 64, 6, 512
    M = B * C * Tp
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(0)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    a = r(M, 256).half(); res = r(M, 256).half()
    wo = r(256, 256, sc=1 / 16).half(); win = r(768, 256, sc=1 / 8).half()
    bo = r(256, sc=0.1); g1 = 1 + r(256, sc=0.1); be1 = r(256, sc=0.1); bin_ = r(768, sc=0.3)
    ws = ops.spk_stream_pack(wo, win)
    x1 = torch.empty_like(res); o1 = torch.empty_like(a)
    x2 = torch.empty_like(res); o2 = torch.empty_like(a); qkv2 = torch.empty(M, 768, dtype=torch.float16, device=dev)

    def new():
        ops.attnout_spk_stream(a, ws, bo, res, g1, be1, 1e-5, x1, bin_, o1, B, C, Tp)

    def old():
        ops.linear_res16_ln(a, wo, bo, res, g1, be1, None, x2, 1e-5)
        ops.linear(x2, win, bin_, qkv2)
        ops.spk_attn(qkv2, o2, B, C, Tp, 4)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    res_ = {"new": [], "old": []}
    for _ in range(5):
        res_["new"].append(timed(new))
        res_["old"].append(timed(old))
    print("max |dx|", (x1.float() - x2.float()).abs().max().item(), "max |do| (valid frames)",
          (o1.float() - o2.float()).view(B * C, Tp, 256)[:, :500].abs().max().item())
    for k, v in res_.items():
        print(f"{k}: min {min(v):.1f} us  median {sorted(v)[len(v) // 2]:.1f} us   {['%.1f' % x for x in v]}")


if __name__ == "__main__":
    main()
