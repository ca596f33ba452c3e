"""One-off stress: the GPU fuzz tests' bodies over many more seeds (not part of the suite)."""
import os, sys, traceback
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import svgrasterize_amd as S
import test_gpu_fuzz as T
bad = 0
lo, hi = int(sys.argv[1]), int(sys.argv[2])
for seed in range(lo, hi):
    for fn in (T.test_random_scene_vs_oracle, T.test_random_affine_per_path_vs_oracle):
        try:
            fn(S, seed)
        except Exception:
            bad += 1
            print("FAIL", fn.__name__, seed); traceback.print_exc(limit=2)
print("seeds", lo, hi, "failures", bad)
