"""bench.py under several option sets: python scripts/gpu_optsweep.py "<common args>" "amg_agg_size=4" "amg_agg_size=6,amg_nu=2" ..."""
import sys, os, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
common = sys.argv[1].split()
for spec in sys.argv[2:]:
    extra = []
    for kv in spec.split(","):
        if kv:
            extra += ["--opt", kv]
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--steps", "3", "--no-f64-rerun"] + common + extra,
                         capture_output=True, text=True)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1])
        c = j["config"]
        print(spec or "default", j["ms_per_step"], c["newton_log_fnorm_kspits_reason"], c["phase_ms_per_step"], c["amg_levels"], flush=True)
    except Exception as e:
        print(spec, "failed", out.stderr[-400:], flush=True)
