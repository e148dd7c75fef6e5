"""env-steps/s of a loop that SYNCHRONISES after every step (an RL loop that reads its observations: policy forward pass between steps) against the
free-running loop bench.py times.  GPU only, diagnostics.   python tools/sync_step_rate.py [B] [steps] [N]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_car_racing_amd.vec_env import VecMultiCarRacing
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2
env = VecMultiCarRacing(B, N, seed=0, auto_reset=True, streams=2)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
pool = torch.rand((64, B, N, 3), device="cuda", generator=g); pool[..., 0] = pool[..., 0] * 2 - 1
st = torch.cuda.current_stream()
def run(sync_every):
    for k in range(100): env.step(pool[k % 64])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps):
        env.step(pool[k % 64])
        if sync_every and k % sync_every == sync_every - 1: st.synchronize()
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0) / 1e6
for rep in range(2):
    print(f"B={B} N={N}: synchronised every step {run(1):.2f} M env-steps/s ({B / run(1) :.1f} us per step), every 4th {run(4):.2f} M, every 16th {run(16):.2f} M, free-running {run(0):.2f} M", flush=True)
env.close()
