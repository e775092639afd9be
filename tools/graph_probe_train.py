"""Study: how much of a training step is launch overhead?  forward + backward eager vs one hipGraph replay of the same launches (the replay
repeats one step's dropout seeds: a timing probe, not a training loop).  usage: python tools/graph_probe_train.py [fs|ls]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
fl = sys.argv[1] if len(sys.argv) > 1 else "fs"
dev = torch.device("cuda:0")
B, T = 64, (1000 if fl == "ls" else 500)
eng, feats, labels = bench.train_setup(dev, B, T, 4, 0, fl)
il = [T] * B
for _ in range(3):
    eng.step(feats, labels, il)
torch.cuda.synchronize()
def fb():
    bf = eng.forward(feats, labels, il)
    eng.backward(bf)
def timed(fn, n=10):
    fn(); torch.cuda.synchronize(); gc.collect(); gc.disable()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); gc.enable()
    return (time.perf_counter() - t0) / n * 1e3
print(f"{fl}: forward + backward eager {timed(fb):.3f} ms", flush=True)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    fb()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    fb()
print(f"{fl}: forward + backward hipGraph replay {timed(g.replay):.3f} ms", flush=True)
print(f"{fl}: full eager step {timed(lambda: eng.step(feats, labels, il)):.3f} ms")
