"""Where does the time of ONE model(x) call go (the evaluate() route: a single stochastic pass on a batch of 250)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_weights_

wl = bench.WORKLOADS["resnet18_me"]
dev = torch.device("cuda", 0)
torch.manual_seed(0); np.random.seed(0)
model = synthetic_weights_(bench._load(wl[0])(**wl[2]), 0).to(dev).eval()
x = synthetic_images(250, seed=1234).to(dev)
for _ in range(3):
    model(x)
torch.cuda.synchronize()
def timed(fn, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("model(x), device-resident input: %.3f ms per call" % timed(lambda: model(x)))
eng = model.engine(dev, max_batch=250)
S = eng.new_moments(250)
print("engine.accumulate(T=1): %.3f ms" % timed(lambda: eng.accumulate(x, S, 0, 1, 0)))
print("engine.new_moments: %.3f ms" % timed(lambda: eng.new_moments(250)))
print("model.advance(1): %.3f ms" % timed(lambda: model.advance(1)))
print("model.mask_cnt0(): %.3f ms" % timed(lambda: model.mask_cnt0()))
print("model.engine(): %.3f ms" % timed(lambda: model.engine(dev, max_batch=250)))
eng.profile(True)
eng.accumulate(x, S, 0, 1, 0); torch.cuda.synchronize()
print({k: (round(v[0], 3), v[1]) for k, v in eng.profile_read().items()})
rows = eng.profile_launches()
eng.profile(False)
for r in sorted(rows, key=lambda r: -r["ms"])[:8]:
    print(r)
xh = synthetic_images(250, seed=1234)
print("H2D of a 250-image batch (pageable): %.3f ms" % timed(lambda: xh.to(dev), 20))
