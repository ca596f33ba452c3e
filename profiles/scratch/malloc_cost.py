import time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import svgrasterize_amd as S
ctx = S.Context.get(0)
ctx.sync()
keep = []
for size in (4096, 1 << 20, 17 << 20, 65 << 20, 130 << 20, 300 << 20, 700 << 20, 4096 + 1, (1 << 20) + 5, (65 << 20) + 3):
    t = time.perf_counter()
    b = ctx.alloc(size)
    dt = time.perf_counter() - t
    keep.append(b)
    print(f"alloc {size/2**20:9.3f} MiB: {dt*1e6:8.1f} us")
