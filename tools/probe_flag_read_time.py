"""When does the HIP runtime read its environment flags (GPU_MAX_HW_QUEUES, AMD_LOG_LEVEL ... all parsed by the same
Flag::init)?  Import torch FIRST, set AMD_LOG_LEVEL afterwards, then touch the device: runtime log lines on stderr mean the
flags are read at the first HIP API call (so a package may still set GPU_MAX_HW_QUEUES at import time, after torch)."""
import os
import sys

import torch

assert not torch.cuda.is_initialized()
os.environ["AMD_LOG_LEVEL"] = "3"
x = torch.zeros(4, device="cuda")
torch.cuda.synchronize()
print("PROBE_DONE initialized=%s" % torch.cuda.is_initialized(), file=sys.stderr)
