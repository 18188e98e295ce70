"""cProfile of one evaluate() pass over the synthetic host loader: where do 33 ms per model(X) call go?"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_labels, synthetic_weights_
from bayesnn_fpga_amd.train.evaluate import MultiExitAccuracy, evaluate

wl = bench.WORKLOADS["resnet18_me"]
dev = torch.device("cuda", 0)
torch.manual_seed(0); np.random.seed(0)
model = synthetic_weights_(bench._load(wl[0])(**wl[2]), 0).to(dev).eval()
x, y = synthetic_images(10000, seed=1234), synthetic_labels(10000, 10, seed=1235)
for pin in (1, 0):
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x, y), batch_size=250, shuffle=False, num_workers=0, pin_memory=bool(pin))
    t0 = time.perf_counter()
    n = sum(1 for _ in loader)
    print(f"pin={pin}: walking the loader alone: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per batch")
    loss = MultiExitAccuracy(4)
    evaluate(loss, loader, model, 0, "x", 1, create_log=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evaluate(loss, loader, model, 0, "x", 2, create_log=False)
    torch.cuda.synchronize()
    print(f"pin={pin}: evaluate: {(time.perf_counter() - t0) / (2 * n) * 1e3:.2f} ms per model call")
pr = cProfile.Profile()
pr.enable()
evaluate(loss, loader, model, 0, "x", 1, create_log=False)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
