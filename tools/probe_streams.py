"""Which pool streams can run beside the engine's compute and side streams without slowing their dispatch?  (volta_amd/streams.py: active_cost)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import volta_amd
from volta_amd import streams as S, _lib as L
import ctypes as C
torch.zeros(1, device="cuda")
own, side = S.engine_streams()
print("GPU_MAX_HW_QUEUES=%s" % os.environ.get("GPU_MAX_HW_QUEUES"))
print("compute vs side: train on compute with side active %.0f / idle %.0f us; train on side with compute active %.0f / %.0f" % (S.active_cost(own, side) + S.active_cost(side, own)))
seen = {}
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    st = torch.cuda.Stream()
    if st.cuda_stream in seen:
        print("pool stream %2d = pool stream %d" % (i, seen[st.cuda_stream]))
        continue
    seen[st.cuda_stream] = i
    a, b = S.active_cost(own, st), S.active_cost(side, st)
    print("pool stream %2d (%#x): launches on compute %.0f us with it active / %.0f idle; on side %.0f / %.0f   %s" % (
        i, st.cuda_stream, a[0], a[1], b[0], b[1], "CLEAN" if a[0] < 1.25 * a[1] and b[0] < 1.25 * b[1] else ""), flush=True)
