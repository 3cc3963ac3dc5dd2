"""Where does the HOST time of one training step go?  (cProfile over a few steps; the GPU work is asynchronous)"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

sys.argv = ["bench.py", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timing"]
# reuse bench's setup by running its main once under the profiler with a few steps
sys.argv = ["bench.py", "--steps", "8", "--warmup", "3", "--no-cpu-baseline", "--no-kernel-timing"]
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("cumulative").print_callees(r"bench.py:\d+\(step\)")
st.print_callees(r"_engine_backward|_engine_forward|clip_grad_norm_|optimization.py:\d+\(step\)|zero_grad|Plan.run|engine.py:\d+\(run\)|prepare_step|bind_inputs|modeling.py:\d+\(forward\)")
print(s.getvalue())
