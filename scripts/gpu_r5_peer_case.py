"""round 5: one case of tests/test_gpu_peer.py by hand (WORLD processes on the one GPU), results printed in full
usage: python scripts/gpu_r5_peer_case.py KIND WORLD ['{"opt": value}']"""
import json, os, socket, subprocess, sys, tempfile
kind, world = sys.argv[1], int(sys.argv[2])
env = dict(os.environ)
if len(sys.argv) > 3: env["SNS_TEST_OPTS"] = sys.argv[3]
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
root = os.environ.get("SNS_TREE", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = tempfile.mkdtemp()
procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "peer_worker.py"), str(r), str(world), str(port), kind, f"{d}/r{r}.json"],
                          stdout=open(f"{d}/r{r}.log", "w"), stderr=subprocess.STDOUT, env=env) for r in range(world)]
for p in procs:
    try: p.wait(timeout=200)
    except subprocess.TimeoutExpired: p.kill()
for r in range(world):
    if os.path.exists(f"{d}/r{r}.json"):
        o = json.load(open(f"{d}/r{r}.json"))
        print(kind, world, env.get("SNS_TEST_OPTS", ""), {k: o.get(k) for k in ("rank", "ok", "stokes_its", "stokes_rnorm", "ksp_its", "err_stokes", "err_newton", "err_spmv", "cycle", "exchanges", "allreduces", "serial", "sns_error", "error")}, flush=True)
    else:
        print("rank", r, "no result:", open(f"{d}/r{r}.log").read()[-1500:])
