"""Single-pair latency of the registration path (configs[1] of BASELINE.json read literally): one pair per call,
no batching.  Run under rocprofv3 --kernel-trace --stats for the per-kernel picture."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eyoc_amd  # noqa: E402
from eyoc_amd import synthetic as syn  # noqa: E402
from eyoc_amd.harness import DeviceBatch, RegistrationConfig, RegistrationPipeline  # noqa: E402

device = torch.device("cuda:0")
cfg = RegistrationConfig()
model = eyoc_amd.load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, conv1_kernel_size=5, normalize_feature=True)
import numpy as np  # noqa: E402
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.make_weights().items()})
model = model.to(device).eval()
if os.environ.get("ST_GROUP"):
    from eyoc_amd import _lib
    _lib.knob("eyoc_spconv_st_group_rows", int(os.environ["ST_GROUP"]))
if os.environ.get("UPC_MIN_ROWS"):
    from eyoc_amd import _lib
    _lib.knob("eyoc_spconv_upc_min_rows", int(os.environ["UPC_MIN_ROWS"]))
if os.environ.get("EYOC_DOWN"):
    from eyoc_amd import _lib
    _lib.knob("eyoc_spconv_select_down_kernel", int(os.environ["EYOC_DOWN"]))
pipe = RegistrationPipeline(model, cfg)
single = DeviceBatch([syn.make_pair(0)], [0], device, cfg.n_points)
for _ in range(3):
    pipe.register(single)
torch.cuda.synchronize()
n = int(os.environ.get("ITERS", "20"))
t0 = time.perf_counter()
for _ in range(n):
    pipe.register(single)
torch.cuda.synchronize()
print(f"single pair: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per call")
