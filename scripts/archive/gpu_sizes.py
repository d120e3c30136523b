"""Iteration counts / step time of bench.py over several duct sizes (single GPU)."""
import sys, os, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for spec in sys.argv[1:]:
    cells, length = spec.split(":")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-f64-rerun", "--steps", "3", "--cells", cells,
                          "--length", length], capture_output=True, text=True)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1])
        c = j["config"]
        print(spec, j["value"], j["ms_per_step"], c["newton_log_fnorm_kspits_reason"], c["phase_ms_per_step"], c["amg_levels"], c["stokes_its"], flush=True)
    except Exception as e:
        print(spec, "failed", out.stderr[-400:], flush=True)
