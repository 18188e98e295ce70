import numpy as np, torch, sys
sys.path.insert(0, ".")
from bayesnn_fpga_amd import _lib
from bayesnn_fpga_amd.models.resnet18.resnet18 import ResNet18MCEarlyExit
from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_weights_
from bayesnn_fpga_amd.train import confidence_exiting as cex
from tests.helpers import build_seeded
DEV = "cuda:0"
for opt in (dict(), dict(conv_pool=0), dict(conv_pool=0, conv_s2=0)):
    for k, v in dict(conv_pool=1, conv_s2=1).items(): _lib.set_option(k, v)
    for k, v in opt.items(): _lib.set_option(k, v)
    kw = dict(dropout_exit=True, dropout="block", mask_type="mask", num_masks=4, mask_scale=4.0, out_dim=10)
    B, T, seed = 45, 6, 11
    m = build_seeded(ResNet18MCEarlyExit, kw); synthetic_weights_(m, 0)
    eng = m.to(DEV).eval().engine(torch.device(DEV), max_batch=B)
    x = synthetic_images(B, seed=21).to(DEV)
    full = eng.predict(x, T, seed=seed)["mean"].cpu().numpy()
    conf = full.max(-1)
    thr = float(np.median(conf[1]))
    r = eng.predict_with_exit(x, T, thr, seed=seed)
    got = r["exit_layer"].cpu().numpy(); mean = r["mean"].cpu().numpy()
    print(opt, "active_after", r["active_after"])
    for e in range(4):
        alive = got >= e
        d = np.abs(mean[e][alive] - full[e][alive]).max(-1)
        idx = np.nonzero(alive)[0]
        bad = [(int(idx[i]), float(d[i])) for i in range(len(d)) if d[i] > 1e-13]
        print(" exit", e, "alive", int(alive.sum()), "bad", bad[:8])
