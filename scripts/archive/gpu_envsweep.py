"""Run bench.py under a few values of one environment variable (experiments only)."""
import sys, os, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
var = sys.argv[1]
for val in sys.argv[2:]:
    env = dict(os.environ); env[var] = val
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-f64-rerun", "--steps", "3"], env=env, capture_output=True, text=True)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1])
        print(var, val, j["ms_per_step"], j["config"]["newton_log_fnorm_kspits_reason"], flush=True)
    except Exception as e:
        print(var, val, "failed", out.stderr[-300:], flush=True)
