import sys, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.dont_write_bytecode = True
import bench
dev = torch.device("cuda")
from fs_eend_amd.fs_model import OnlineTransformerDADiarization
torch.manual_seed(0)
model = OnlineTransformerDADiarization(n_speakers=None, in_size=345, **bench.FS_CFG).eval().to(dev)
C = 6
print("A: (500,) alone", bench.length_sweep(dev, model, C, lengths=(500,)))
print("B: (300, 500)", bench.length_sweep(dev, model, C, lengths=(300, 500)))
model._ws.clear(); torch.cuda.empty_cache()
print("C: after clear (500,)", bench.length_sweep(dev, model, C, lengths=(500,)))
# eager per-op timing at the sweep's inputs
T = 500; Bs = 64
g = torch.Generator().manual_seed(977 + T)
src = [(torch.randn(T, 345, generator=g) * 2 - 3).to(dev) for _ in range(Bs)]
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): model.test(src, [T] * Bs, C)
    torch.cuda.synchronize(); print("eager ms/step", (time.perf_counter() - t0) / 5 * 1e3)
g = torch.Generator().manual_seed(1)
src2 = [torch.randn(T, 345, generator=g).to(dev) for _ in range(Bs)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): model.test(src2, [T] * Bs, C)
torch.cuda.synchronize(); print("eager ms/step (plain randn input)", (time.perf_counter() - t0) / 5 * 1e3)
