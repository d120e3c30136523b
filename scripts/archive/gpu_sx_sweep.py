"""Team-transport rehearsal (8 thin slabs) under several option sets, e.g. amg_replicate_rows=0 amg_replicate_rows=65536."""
import sys, os, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = sys.argv[1] if len(sys.argv) > 1 else "150,38,38"
for thr in sys.argv[2:]:
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "gpu_weak_rehearsal.py"), "8", base,
                          thr], capture_output=True, text=True)
    lines = [l for l in out.stdout.splitlines() if "stokes its" in l or l.startswith("N=")]
    print("thr", thr, lines[0] if lines else out.stderr[-300:], "|", lines[1].strip() if len(lines) > 1 else "", flush=True)
