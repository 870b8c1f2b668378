import torch

DEVICE_TYPE = "cuda" if torch.cuda.is_available() else "cpu"   # ROCm devices are `cuda` in torch
DIST_BACKEND = "nccl" if DEVICE_TYPE == "cuda" else "gloo"      # nccl IS RCCL on ROCm
