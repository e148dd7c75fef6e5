"""Double-buffered stepping: two handles of B envs each on streams of their own (A steps while B's observations are consumed) against one
handle of B and one of 2B envs.  GPU only, diagnostics.   python tools/two_handles_rate.py [B] [steps]"""
import sys, os, time, warnings, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_car_racing_amd.vec_env import VecMultiCarRacing
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
N = 2
def pool_of(b):
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    p = torch.rand((64, b, N, 3), device="cuda", generator=g); p[..., 0] = p[..., 0] * 2 - 1
    return p
def one(b):
    env = VecMultiCarRacing(b, N, seed=0, auto_reset=True, streams=2); env.reset(); pool = pool_of(b)
    for k in range(100): env.step(pool[k % 64])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps):
        env.step(pool[k % 64])
        if k % 16 == 15: torch.cuda.current_stream().synchronize()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    o = int(env.L.mcr_step_ordering(env.h)); env.close()
    return b * steps / dt, o
def two(b):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ea = VecMultiCarRacing(b, N, seed=0, auto_reset=True, streams=2); eb = VecMultiCarRacing(b, N, seed=0, env_offset=b, auto_reset=True, streams=2)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(sa): ea.reset()
    with torch.cuda.stream(sb): eb.reset()
    pool = pool_of(b)
    def loop(n):
        for k in range(n):
            with torch.cuda.stream(sa): ea.step(pool[k % 64])
            with torch.cuda.stream(sb): eb.step(pool[(k + 7) % 64])
            if k % 16 == 15: sa.synchronize(); sb.synchronize()
    loop(100); torch.cuda.synchronize(); t0 = time.perf_counter(); loop(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    o = (int(ea.L.mcr_step_ordering(ea.h)), int(eb.L.mcr_step_ordering(eb.h)))
    st = (ea.status_words().tolist(), eb.status_words().tolist())
    ea.close(); eb.close()
    return 2 * b * steps / dt, o, st
r1, o1 = one(B); r2, o2 = one(2 * B); r3, o3, st = two(B)
print(f"one handle x {B}: {r1 / 1e6:.2f} M env-steps/s (ordering {o1}); one handle x {2 * B}: {r2 / 1e6:.2f} M (ordering {o2}); two handles x {B} on two streams: {r3 / 1e6:.2f} M (orderings {o3}, status {st})")
