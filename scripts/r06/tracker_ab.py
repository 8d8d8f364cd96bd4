"""bench.py's tracker_call for several builds, alternating (child process per library):  python scripts/r06/tracker_ab.py lib1.so lib2.so ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    import torch, bench
    r = bench.tracker_call(torch.device("cuda:0"), reps=60)
    print(json.dumps({"lib": os.path.basename(sys.argv[2]), **{k: r[k] for k in ("ms_per_call", "ms_per_call_min", "ms_solve", "ms_cache", "ms_upload", "ms_total")}}))
else:
    for lib in sys.argv[1:]:
        env = dict(os.environ, BTBA_LIB_PATH=os.path.abspath(lib))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib], env=env, capture_output=True, text=True, timeout=300)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-400:], flush=True)
