import sys, os
sys.path.insert(0, os.getcwd())
import torch
from bayesnn_fpga_amd.models.resnet18.resnet18 import ResNet18MCEarlyExit
from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_weights_
kw = dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10)
torch.manual_seed(0)
model = synthetic_weights_(ResNet18MCEarlyExit(**kw), 0).to("cuda:0").eval()
x = synthetic_images(4, seed=1234)
eng = model.engine(torch.device("cuda:0"), max_batch=4)
print("chunk", eng.chunk_samples, "ws", eng.workspace_bytes)
try:
    r = eng.predict(x.to("cuda:0"), 4, seed=42)
    torch.cuda.synchronize()
    print("ok", float(r["mean"].sum()))
except Exception as e:
    print("FAIL", e)
