"""Timing of render('rgb_array') (mcr_render, 600x400 per agent): frames/s of one env. GPU only."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_car_racing_amd.vec_env import VecMultiCarRacing
for N in (1, 2, 8):
    env = VecMultiCarRacing(4, N, seed=0, auto_reset=True); env.reset()
    a = torch.rand((4, N, 3), device="cuda")
    for _ in range(60): env.step(a)
    for _ in range(5): env.render_rgb(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): out = env.render_rgb(0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 200
    byt = N * 600 * 400 * 3
    print(f"N={N}: {ms * 1e3:.1f} us per render call ({N} frames of 600x400), {byt / ms / 1e6:.1f} GB/s of frame bytes")
    env.close()
