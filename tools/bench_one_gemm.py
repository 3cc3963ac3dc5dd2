"""One GEMM shape, a few launches: target for rocprofv3 --pmc runs.  usage: bench_one_gemm.py LAYOUT M N K [tile]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import torch
from volta_amd import _lib as L, ops
import bench_gemm
lay = {"NT": L.NT, "NN": L.NN, "TN": L.TN}[sys.argv[1]]
M, N, K = (int(x) for x in sys.argv[2:5])
if len(sys.argv) > 5:
    L.lib.vk_gemm_set_tile.argtypes = [ctypes.c_int]
    L.lib.vk_gemm_set_tile(int(sys.argv[5]))
bench_gemm.bench("%s %dx%dx%d" % (sys.argv[1], M, N, K), lay, L.EPI_F32 if lay == L.TN else L.EPI_BF16, M, N, K, iters=10)
