import os, sys, ctypes
os.environ.setdefault("GPU_MAX_HW_QUEUES","8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import _lib as L
import numpy as np
lib=L.product()
lib.ebcc_hip_workspace_bytes.restype=ctypes.c_size_t
free0,total=torch.cuda.mem_get_info()
with L.Context(256,721,1440) as ctx:
    print("engine for 256 frames: workspace GB", lib.ebcc_hip_workspace_bytes(ctypes.c_void_p(ctx.ptr))/1e9)
    frames=np.stack([L.era5_like(721,1440,s) for s in range(4)]*64)
    cfg=L.make_config((1,721,1440),base_cr=30.0,error=0.5,residual_type=L.MAX_ERROR)
    got=ctx.encode_frames(frames,cfg); dec=ctx.decode_frames(got)
    free1,_=torch.cuda.mem_get_info()
    print("device memory in use after an encode+decode of 256 frames: GB", (free0-free1)/1e9, "of", total/1e9)
